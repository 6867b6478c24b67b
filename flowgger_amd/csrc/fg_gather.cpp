// fg_gather.cpp -- the ordered HOST gather of the multi-GPU path as part of the C ABI.
//
// Lines decode independently, so a batch is cut into contiguous line ranges (fg_shard_plan), one per GPU; each rank
// decodes its range into its own tables.  What the reference guarantees downstream is ORDER: handle_line runs for the
// lines of a connection in input order (src/flowgger/splitter/line_splitter.rs:17-54).  These functions put the shard
// tables back into one table in that order -- plain memory moves over (pinned) host buffers, no Python, no per-line
// dispatch, threaded over row ranges:
//   fg_gather_tables   shards of ONE batch, in shard order (concatenation; entry indices rebased, spans are
//                      line-relative and stay as they are)
//   fg_merge_tables    BASELINE configuration 5: sub-batches that were split off by FORMAT (the reference has one
//                      decoder per input, flowgger/mod.rs:413-422, so "mixed RFC5424 + LTSV" is two tagged sub-batches)
//                      go back to their original positions by line index
//   fg_ordered_merge   the same for byte records of variable size (canonical Record blobs, encoded messages)
// Host-only code: no HIP dependency.
#include <algorithm>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/fg_hip.h"

namespace {

unsigned pool_size(uint64_t work_bytes) {
    if (work_bytes < (8u << 20)) return 1;  // small batches: thread start-up costs more than the copy
    unsigned hw = std::thread::hardware_concurrency();
    if (hw == 0) hw = 1;
    return std::min(hw, 16u);
}

template <class F>
void parallel_for(uint64_t n, unsigned threads, F f) {
    if (threads <= 1 || n < 2) {
        f(0, n);
        return;
    }
    std::vector<std::thread> th;
    th.reserve(threads);
    for (unsigned t = 0; t < threads; ++t) {
        const uint64_t a = n * t / threads, b = n * (t + 1) / threads;
        if (b > a) th.emplace_back([=] { f(a, b); });
    }
    for (auto& x : th) x.join();
}

uint64_t used_of(const fg_tables& t) { return t.ent_used ? *t.ent_used : 0; }

bool part_ok(const fg_tables& p) {
    if (p.n && (!p.meta || !p.ts || !p.hostname || !p.appname || !p.procid || !p.msgid || !p.msg || !p.full_msg || !p.ent_first ||
                !p.ent_count))
        return false;
    const uint64_t u = used_of(p);
    if (u > p.ent_cap) return false;
    if (u && (!p.ent_name || !p.ent_val || !p.ent_type || !p.ent_flags)) return false;
    return true;
}

// the output table of a gather / merge: every column the n rows and e entries are written to must be there
bool out_ok(const fg_tables* out, uint64_t n, uint64_t e) {
    if (!out || out->n < n || out->ent_cap < e || !out->ent_used) return false;
    if (n && (!out->meta || !out->ts || !out->hostname || !out->appname || !out->procid || !out->msgid || !out->msg || !out->full_msg ||
              !out->ent_first || !out->ent_count))
        return false;
    if (e && (!out->ent_name || !out->ent_val || !out->ent_type || !out->ent_flags)) return false;
    return true;
}

// entry columns of part k -> out[base ..]
void copy_entries(const fg_tables& p, uint64_t base, fg_tables* out) {
    const uint64_t u = used_of(p);
    if (!u) return;
    memcpy(out->ent_name + base, p.ent_name, u * sizeof(fg_span));
    memcpy(out->ent_val + base, p.ent_val, u * 8);
    memcpy(out->ent_type + base, p.ent_type, u);
    memcpy(out->ent_flags + base, p.ent_flags, u);
}

}  // namespace

extern "C" {

int fg_gather_size(const fg_tables* parts, uint32_t g, uint64_t* n_rows, uint64_t* n_entries) {
    if ((g && !parts) || !n_rows || !n_entries) return FG_ERR_ARG;
    uint64_t n = 0, e = 0;
    for (uint32_t k = 0; k < g; ++k) {
        if (!part_ok(parts[k])) return FG_ERR_ARG;
        n += parts[k].n;
        e += used_of(parts[k]);
    }
    *n_rows = n;
    *n_entries = e;
    return FG_OK;
}

int fg_gather_tables(const fg_tables* parts, uint32_t g, fg_tables* out) {
    uint64_t n = 0, e = 0;
    int rc = fg_gather_size(parts, g, &n, &e);
    if (rc != FG_OK) return rc;
    if (!out_ok(out, n, e)) return FG_ERR_ARG;
    if (e > 0xFFFFFFFFull) return FG_ERR_ENT_OVERFLOW;  // ent_first is 32 bits wide
    uint64_t row0 = 0, ent0 = 0;
    uint64_t bytes = n * FG_ROW_BYTES + e * FG_ENT_BYTES;
    const unsigned threads = pool_size(bytes);
    for (uint32_t k = 0; k < g; ++k) {
        const fg_tables& p = parts[k];
        const uint64_t r0 = row0, b = ent0;
        parallel_for(p.n, threads, [&, r0, b](uint64_t a, uint64_t z) {
            const uint64_t m = z - a;
            memcpy(out->meta + r0 + a, p.meta + a, m * 4);
            memcpy(out->ts + r0 + a, p.ts + a, m * 8);
            fg_span* const dst[6] = {out->hostname, out->appname, out->procid, out->msgid, out->msg, out->full_msg};
            const fg_span* const src[6] = {p.hostname, p.appname, p.procid, p.msgid, p.msg, p.full_msg};
            for (int j = 0; j < 6; ++j) memcpy(dst[j] + r0 + a, src[j] + a, m * sizeof(fg_span));
            memcpy(out->ent_count + r0 + a, p.ent_count + a, m * 4);
            for (uint64_t i = a; i < z; ++i) out->ent_first[r0 + i] = p.ent_count[i] ? (uint32_t)(p.ent_first[i] + b) : 0u;
        });
        copy_entries(p, b, out);
        row0 += p.n;
        ent0 += used_of(p);
    }
    *out->ent_used = e;
    return FG_OK;
}

int fg_merge_tables(const fg_tables* parts, uint32_t g, const uint64_t* const* index, fg_tables* out, uint8_t* src_part) {
    uint64_t n = 0, e = 0;
    int rc = fg_gather_size(parts, g, &n, &e);
    if (rc != FG_OK) return rc;
    if (!out_ok(out, n, e) || (g && !index)) return FG_ERR_ARG;
    if (e > 0xFFFFFFFFull) return FG_ERR_ENT_OVERFLOW;
    if (g > 255) return FG_ERR_ARG;
    // every original position exactly once, increasing inside a part (sub-batches keep their relative order)
    {
        std::vector<uint8_t> seen(n, 0);
        for (uint32_t k = 0; k < g; ++k) {
            if (parts[k].n && !index[k]) return FG_ERR_ARG;
            for (uint64_t j = 0; j < parts[k].n; ++j) {
                const uint64_t i = index[k][j];
                if (i >= n || seen[i] || (j && index[k][j - 1] >= i)) return FG_ERR_ARG;
                seen[i] = 1;
            }
        }
    }
    uint64_t ent0 = 0;
    const unsigned threads = pool_size(n * FG_ROW_BYTES + e * FG_ENT_BYTES);
    for (uint32_t k = 0; k < g; ++k) {
        const fg_tables& p = parts[k];
        const uint64_t b = ent0;
        const uint64_t* ix = index[k];
        parallel_for(p.n, threads, [&, b, ix, k](uint64_t a, uint64_t z) {
            fg_span* const dst[6] = {out->hostname, out->appname, out->procid, out->msgid, out->msg, out->full_msg};
            const fg_span* const src[6] = {p.hostname, p.appname, p.procid, p.msgid, p.msg, p.full_msg};
            for (uint64_t j = a; j < z; ++j) {
                const uint64_t i = ix[j];
                out->meta[i] = p.meta[j];
                out->ts[i] = p.ts[j];
                for (int c = 0; c < 6; ++c) dst[c][i] = src[c][j];
                out->ent_count[i] = p.ent_count[j];
                out->ent_first[i] = p.ent_count[j] ? (uint32_t)(p.ent_first[j] + b) : 0u;
                if (src_part) src_part[i] = (uint8_t)k;
            }
        });
        copy_entries(p, b, out);
        ent0 += used_of(p);
    }
    *out->ent_used = e;
    return FG_OK;
}

int64_t fg_ordered_merge(uint32_t g, const uint64_t* m, const uint64_t* const* index, const uint8_t* const* blobs,
                         const uint64_t* const* offs, uint8_t* out, uint64_t cap, uint64_t* out_offs) {
    if ((g && (!m || !index || !blobs || !offs)) || !out_offs) return FG_ERR_ARG;
    uint64_t n = 0;
    for (uint32_t k = 0; k < g; ++k) n += m[k];
    // pass 1: sizes at the original positions -> exclusive scan
    for (uint64_t i = 0; i <= n; ++i) out_offs[i] = ~0ull;
    for (uint32_t k = 0; k < g; ++k) {
        if (m[k] && (!index[k] || !offs[k] || !blobs[k])) return FG_ERR_ARG;
        for (uint64_t j = 0; j < m[k]; ++j) {
            const uint64_t i = index[k][j];
            if (i >= n || out_offs[i + 1] != ~0ull || offs[k][j + 1] < offs[k][j] || (j && index[k][j - 1] >= i)) return FG_ERR_ARG;
            out_offs[i + 1] = offs[k][j + 1] - offs[k][j];
        }
    }
    out_offs[0] = 0;
    for (uint64_t i = 0; i < n; ++i) out_offs[i + 1] += out_offs[i];
    const uint64_t total = out_offs[n];
    if (!out || total > cap) return (int64_t)total;  // sizing call
    const unsigned threads = pool_size(total);
    for (uint32_t k = 0; k < g; ++k) {
        const uint64_t* ix = index[k];
        const uint64_t* o = offs[k];
        const uint8_t* src = blobs[k];
        parallel_for(m[k], threads, [=](uint64_t a, uint64_t z) {
            // a run of lines that were neighbours in the original order is one contiguous copy
            uint64_t j = a;
            while (j < z) {
                uint64_t r = j + 1;
                while (r < z && ix[r] == ix[r - 1] + 1) ++r;
                memcpy(out + out_offs[ix[j]], src + o[j], o[r] - o[j]);
                j = r;
            }
        });
    }
    return (int64_t)total;
}

}  // extern "C"
