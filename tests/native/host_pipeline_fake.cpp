// The HOST side of the C ABI (flowgger_amd/csrc/fg_capi.cpp + fg_host_pipeline.cpp + fg_gather.cpp + fg_materialize.cpp, unchanged) compiled with g++ against
// a synchronous stand-in for the HIP runtime (tests/native/fakehip) and the FAKE kernel launchers below, so that `pytest -m "not gpu"`
// runs its bookkeeping: slices cut at line boundaries, rows at their final index, entry ranges per slice of one shared counter, the
// retry when the entry table is too small, the raw-stream path's frame counts per slice and its fall-back, error paths.
// The fake decode "kernel" is deterministic nonsense with the REAL kernels' contract: one row per line, one entry per '=' of the
// line, slots taken from the shared counter (which keeps counting past the capacity), FG_ST_OVERFLOW rows without entries,
// terminators stripped as the kernels strip them, FG_ST_BAD_UTF8 for flagged frames.  The fake framing "kernels" are a plain
// delimiter scan with the real ones' contract (ranks continue from slice to slice; a frame is bad when it holds a byte >= 0xF8).
// Test infrastructure: nothing here is shipped, and it says nothing about what the GPU computes.
#include "../../flowgger_amd/csrc/fg_capi.cpp"
#include "../../flowgger_amd/csrc/fg_host_pipeline.cpp"
#include "../../flowgger_amd/csrc/fg_gather.cpp"
#include "../../flowgger_amd/csrc/fg_materialize.cpp"

namespace {
unsigned long long g_launches = 0;
unsigned long long g_kernel_uploaded = 0;  // bytes the fake framing scan copied itself (the raw-stream path without the copy engine)

int fake_decode(const uint8_t* bytes, const uint64_t* offsets, uint64_t n, const fg::DevTables* t, uint32_t strip, const uint8_t* line_bad) {
    ++g_launches;
    const fg_span none{0u, FG_NONE};
    for (uint64_t i = 0; i < n; ++i) {
        const uint8_t* ln = bytes + offsets[i];
        uint32_t len = (uint32_t)(offsets[i + 1] - offsets[i]);
        if (strip == FG_FRAME_LINE && len && ln[len - 1] == '\n') {
            --len;
            if (len && ln[len - 1] == '\r') --len;
        } else if (strip == FG_FRAME_NUL && len && ln[len - 1] == 0) {
            --len;
        }
        uint32_t status = len == 0 ? 1u : 0u;
        if (line_bad && line_bad[i]) status = FG_ST_BAD_UTF8;
        uint32_t k = 0;
        if (status == 0)
            for (uint32_t p = 0; p < len; ++p) k += ln[p] == '=';
        uint32_t first = 0;
        if (k) {
            const unsigned long long at = *t->ent_used;
            *t->ent_used = at + k;  // (keeps counting: the final value is what the batch needs)
            if (at + k > t->ent_cap) {
                status = FG_ST_OVERFLOW;
                k = 0;
            } else {
                first = (uint32_t)at;
                uint32_t j = 0;
                for (uint32_t p = 0; p < len; ++p)
                    if (ln[p] == '=') {
                        t->ent_name[first + j] = fg_span{p ? p - 1u : 0u, 1u};
                        const uint32_t vl = len - p - 1u < 3u ? len - p - 1u : 3u;
                        t->ent_val[first + j] = (uint64_t)(p + 1u) | ((uint64_t)vl << 32);
                        t->ent_type[first + j] = FG_T_STRING;
                        t->ent_flags[first + j] = 0;
                        ++j;
                    }
            }
        }
        const bool ok = status == 0;
        t->meta[i] = status | ((ok ? (uint32_t)(ln[0] & 0x17u) : 0xFFu) << 8) | ((ok ? len & 7u : 0xFFu) << 16);
        t->ts[i] = ok ? (double)len + 0.5 : 0.0;
        t->span[0][i] = ok ? fg_span{0u, len < 4u ? len : 4u} : none;
        t->span[1][i] = none;
        t->span[2][i] = none;
        t->span[3][i] = none;
        t->span[4][i] = ok && len > 1u ? fg_span{len / 2u, len - len / 2u} : none;
        t->span[5][i] = ok ? fg_span{0u, len} : none;
        t->ent_first[i] = first;
        t->ent_count[i] = k;
    }
    return 0;
}
}  // namespace

extern "C" unsigned long long fgf_kernel_uploaded(int reset) {
    const unsigned long long v = g_kernel_uploaded;
    if (reset) g_kernel_uploaded = 0;
    return v;
}
extern "C" unsigned long long fgf_launches(int reset) {
    const unsigned long long v = g_launches;
    if (reset) g_launches = 0;
    return v;
}
extern "C" void fgf_counters(unsigned long long out[3], int reset) {
    out[0] = fakehip::copies();
    out[1] = fakehip::bytes_h2d();
    out[2] = fakehip::bytes_d2h();
    if (reset) fakehip::copies() = fakehip::bytes_h2d() = fakehip::bytes_d2h() = 0;
}
extern "C" void fgf_fail_malloc_after(long long n) { fakehip::fail_malloc_after() = n; }

extern "C" uint64_t fg_stash_bytes(uint32_t blocks) { return 64ull * blocks; }
extern "C" int fg_launch_poke64(const uint64_t* src, uint64_t* dst, hipStream_t) {
    *dst = *src;
    return 0;
}
extern "C" int fg_launch_calib(int mode, const uint8_t* src, uint8_t* dst, uint64_t nbytes, uint32_t*, hipStream_t) {
    if (mode != 1) memcpy(dst, src, nbytes / 16 * 16);
    return 0;
}
extern "C" int fg_launch_rfc5424(const uint8_t* b, const uint64_t* o, uint64_t n, const fg::DevTables* t, uint64_t, hipStream_t, uint64_t*, uint32_t,
                                 uint32_t strip, const uint8_t* bad, const fg_launch_opts*, fg::TicketSlot*) { return fake_decode(b, o, n, t, strip, bad); }
extern "C" int fg_launch_ltsv(const uint8_t* b, const uint64_t* o, uint64_t n, const fg::DevTables* t, const fg::LtsvDevCfg*, uint64_t, hipStream_t,
                              uint64_t*, uint32_t, uint32_t strip, const uint8_t* bad, const fg_launch_opts*, fg::TicketSlot*) { return fake_decode(b, o, n, t, strip, bad); }
extern "C" int fg_launch_gelf_general(const uint8_t*, const uint64_t*, uint64_t, const fg::DevTables*, hipStream_t, uint32_t, const uint8_t*) {
    return 0;  // (the fake fast form leaves nothing pending)
}
extern "C" int fg_launch_gelf(const uint8_t* b, const uint64_t* o, uint64_t n, const fg::DevTables* t, uint64_t, hipStream_t, uint64_t*, uint32_t,
                              uint32_t strip, const uint8_t* bad, const fg_launch_opts*, fg::TicketSlot*) { return fake_decode(b, o, n, t, strip, bad); }
extern "C" int fg_launch_rfc3164(const uint8_t* b, const uint64_t* o, uint64_t n, const fg::DevTables* t, const fg::r3164::Cfg*, uint32_t, hipStream_t,
                                 uint32_t strip, const uint8_t* bad, uint8_t*, int) { return fake_decode(b, o, n, t, strip, bad); }
extern "C" uint64_t fg_rfc3164_scratch_bytes(uint64_t n) { return n * 17 + 4096; }
extern "C" uint64_t fg_rfc3164_regroup_from(void) { return 1u << 20; }
// ---- the fake encoder: an Ok row's message = its line + one '#' per entry + '\n' (so rows AND entry counts of the right slice matter);
//      other rows encode to nothing with enc_status 1.  Contracts of the real launchers (fg_encode.hip): count -> sizes per line +
//      sums per 64 lines; scan -> out_offsets[0 .. n] absolute from `base`; sizes = both with base 0; write -> the bytes.
static uint32_t fake_size(const uint64_t* o, const fg::DevTables* t, uint64_t i) {
    return (t->meta[i] & 0xFFu) == 0u ? (uint32_t)(o[i + 1] - o[i]) + t->ent_count[i] + 1u : 0u;
}
extern "C" int fg_launch_encode_count(const uint8_t*, const uint64_t* o, uint64_t n, const fg::DevTables* t, const fg::EncCfg*, uint32_t, uint32_t,
                                      uint32_t* d_sizes, uint64_t* d_block_sums, uint8_t* d_status, hipStream_t) {
    for (uint64_t i = 0; i < n; ++i) {
        d_sizes[i] = fake_size(o, t, i);
        d_status[i] = (t->meta[i] & 0xFFu) == 0u ? 0 : 1;
        if (i % 64 == 0) d_block_sums[i / 64] = 0;
        d_block_sums[i / 64] += d_sizes[i];
    }
    return 0;
}
extern "C" int fg_launch_encode_scan(const uint32_t* d_sizes, uint64_t* d_block_sums, uint64_t n, uint64_t* d_out_offsets, uint64_t base, hipStream_t) {
    uint64_t at = base, check = 0;
    for (uint64_t i = 0; i < n; ++i) {
        d_out_offsets[i] = at;
        at += d_sizes[i];
    }
    d_out_offsets[n] = at;
    for (uint64_t j = 0; j < (n + 63) / 64; ++j) check += d_block_sums[j];  // (the sums the count step left must be this slice's)
    return check == at - base ? 0 : -7;
}
extern "C" int fg_launch_encode_sizes(const uint8_t* b, const uint64_t* o, uint64_t n, const fg::DevTables* t, const fg::EncCfg* c, uint32_t tc, uint32_t cl,
                                      uint32_t* d_sizes, uint64_t* d_block_sums, uint8_t* d_status, uint64_t* d_out_offsets, hipStream_t s) {
    const int rc = fg_launch_encode_count(b, o, n, t, c, tc, cl, d_sizes, d_block_sums, d_status, s);
    return rc ? rc : fg_launch_encode_scan(d_sizes, d_block_sums, n, d_out_offsets, 0, s);
}
extern "C" int fg_launch_encode_write(const uint8_t* b, const uint64_t* o, uint64_t n, const fg::DevTables* t, const fg::EncCfg*, uint32_t, uint32_t,
                                      const uint64_t* d_out_offsets, uint8_t* d_out, const uint32_t* d_sizes, hipStream_t) {
    for (uint64_t i = 0; i < n; ++i) {
        if ((t->meta[i] & 0xFFu) != 0u) continue;
        if (!d_sizes || (d_sizes[i] & 0x7FFFFFFFu) != fake_size(o, t, i)) return -8;  // (the write step is handed the count step's sizes of ITS slice)
        uint8_t* w = d_out + d_out_offsets[i];
        const uint32_t len = (uint32_t)(o[i + 1] - o[i]);
        memcpy(w, b + o[i], len);
        memset(w + len, '#', t->ent_count[i]);
        w[len + t->ent_count[i]] = '\n';
    }
    return 0;
}
// ---- the device merge (fg_merge.hip), on "device" memory that is host memory here
extern "C" uint64_t fg_merge_scratch_bytes(uint64_t rows) { return rows + 64; }
extern "C" int fg_launch_merge_device(const fg_tables* parts, uint32_t g, const uint64_t* const* d_index, const fg_tables* out, uint8_t* d_src_part,
                                      uint64_t, uint8_t* scratch, hipStream_t) {
    // the real kernels' contract: rows at their arrival positions, entries DENSE in arrival order
    uint8_t* src = d_src_part ? d_src_part : scratch;
    std::vector<uint32_t> local(out->n, 0);
    for (uint64_t i = 0; i < out->n; ++i) out->ent_count[i] = 0;
    for (uint32_t k = 0; k < g; ++k) {
        const fg_tables& p = parts[k];
        const uint64_t used = *p.ent_used < p.ent_cap ? *p.ent_used : p.ent_cap;
        for (uint64_t j = 0; j < p.n; ++j) {
            const uint64_t i = d_index[k][j];
            out->meta[i] = p.meta[j]; out->ts[i] = p.ts[j];
            out->hostname[i] = p.hostname[j]; out->appname[i] = p.appname[j]; out->procid[i] = p.procid[j];
            out->msgid[i] = p.msgid[j]; out->msg[i] = p.msg[j]; out->full_msg[i] = p.full_msg[j];
            out->ent_count[i] = (uint64_t)p.ent_first[j] + p.ent_count[j] <= used ? p.ent_count[j] : 0u;
            local[i] = p.ent_first[j];
            src[i] = (uint8_t)k;
        }
    }
    uint64_t at = 0;
    for (uint64_t i = 0; i < out->n; ++i) {
        const fg_tables& p = parts[src[i]];
        for (uint32_t e = 0; e < out->ent_count[i]; ++e) {
            out->ent_name[at + e] = p.ent_name[local[i] + e]; out->ent_val[at + e] = p.ent_val[local[i] + e];
            out->ent_type[at + e] = p.ent_type[local[i] + e]; out->ent_flags[at + e] = p.ent_flags[local[i] + e];
        }
        out->ent_first[i] = out->ent_count[i] ? (uint32_t)at : 0u;
        at += out->ent_count[i];
    }
    *out->ent_used = at;
    return 0;
}

// ---- framing: offsets[0] = 0, offsets[r + 1] = the byte behind the delimiter of rank r; the word behind block k of `scratch` holds
//      the delimiters of the blocks before k (so the word at [blk1] is "up to the end of this slice")
extern "C" uint64_t fg_frame_block_bytes(void) { return 16384; }
extern "C" uint64_t fg_frame_slice_align(void) { return 131072; }
extern "C" uint64_t fg_frame_scratch_bytes(uint64_t nbytes) { return (nbytes / 16384 + 3) * 8; }
static int frame_blocks_fake(const uint8_t* d_bytes, uint64_t nbytes, uint32_t delim, uint8_t* scratch, uint64_t* d_offsets, uint8_t* d_bad, uint64_t cap,
                             uint64_t blk0, uint64_t blk1, uint64_t** d_total_out) {
    uint64_t* pref = reinterpret_cast<uint64_t*>(scratch);
    if (blk0 == 0) {
        pref[0] = 0;
        d_offsets[0] = 0;
    }
    uint64_t rank = pref[blk0];
    const uint64_t b0 = blk0 * 16384, b1 = blk1 * 16384 < nbytes ? blk1 * 16384 : nbytes;
    for (uint64_t p = b0; p < b1; ++p) {
        if (d_bytes[p] >= 0xF8u && rank < cap) d_bad[rank] = 1;  // (the frame this byte belongs to has rank = delimiters before it)
        if (d_bytes[p] == (uint8_t)delim) {
            if (rank + 1 <= cap + 1) d_offsets[rank + 1] = p + 1;
            ++rank;
        }
    }
    for (uint64_t k = blk0 + 1; k <= blk1; ++k) pref[k] = rank;  // (only [blk1] is read)
    *d_total_out = pref + blk1;
    return 0;
}
// The one-pass scan's look-back giving up (FG_FRAME_ABORTED in the total word): g_frame_abort_left one-pass launches report it; the host
// must come back with classic = 1 (one-piece) or leave the sliced path for the one-piece one.
int g_frame_abort_left = 0;
int g_frame_classic_launches = 0;
extern "C" void fake_frame_abort_next(int n) { g_frame_abort_left = n; g_frame_classic_launches = 0; }
extern "C" int fake_frame_classic_launches(void) { return g_frame_classic_launches; }
static bool fake_aborts(int classic, uint8_t* scratch, uint64_t blk1, uint64_t** d_total_out) {
    if (classic) {
        ++g_frame_classic_launches;
        return false;
    }
    if (g_frame_abort_left <= 0) return false;
    --g_frame_abort_left;
    uint64_t* pref = reinterpret_cast<uint64_t*>(scratch);
    pref[blk1] = ~0ull;
    *d_total_out = pref + blk1;
    return true;
}
extern "C" int fg_launch_frame(const uint8_t* d_bytes, uint64_t nbytes, uint32_t delim, uint8_t* scratch, uint64_t* d_offsets, uint8_t* d_bad, uint64_t cap,
                               uint64_t** d_total_out, hipStream_t, int classic) {
    memset(d_bad, 0, cap);
    if (fake_aborts(classic, scratch, nbytes / 16384 + 1, d_total_out)) return 0;
    const int rc = frame_blocks_fake(d_bytes, nbytes, delim, scratch, d_offsets, d_bad, cap, 0, nbytes / 16384 + 1, d_total_out);
    const uint64_t total = **d_total_out;  // the whole-stream form also ends an unterminated last frame at nbytes
    if (total + 1 <= cap && d_offsets[total] != nbytes) d_offsets[total + 1] = nbytes;
    return rc;
}
extern "C" int fg_launch_frame_slice(const uint8_t* d_bytes, uint64_t nbytes, uint32_t delim, uint8_t* scratch, uint64_t* d_offsets, uint8_t* d_bad,
                                     uint64_t cap, uint64_t blk0, uint64_t blk1, uint64_t** d_total_out, hipStream_t, const uint8_t* src, int classic) {
    if (blk1 > nbytes / 16384 + 1 || blk0 >= blk1) return -1;
    if (fake_aborts(classic, scratch, blk1, d_total_out)) return 0;
    if (src) {  // the scan uploads its blocks itself (whole 16-byte chunks, as the kernel's bounded buffer stores do)
        const uint64_t b0 = blk0 * 16384, b1 = blk1 * 16384 < nbytes ? blk1 * 16384 : nbytes;
        if (b1 > b0) memcpy(const_cast<uint8_t*>(d_bytes) + b0, src + b0, b1 - b0);
        g_kernel_uploaded += b1 > b0 ? b1 - b0 : 0;
    }
    return frame_blocks_fake(d_bytes, nbytes, delim, scratch, d_offsets, d_bad, cap, blk0, blk1, d_total_out);
}

// ---- the fused launches (fg_fused.hpp): frame + decode of a raw chunk in one "kernel".  The real contract: offsets[i] = start of frame
//      i, offsets[frames] = end of the last frame; rows and offsets only below `cap`; an unterminated tail is a frame iff `final`;
//      the two result words = frames (may exceed cap), abort flag.
int g_fused_abort_left = 0;
extern "C" void fake_fused_abort_next(int n) { g_fused_abort_left = n; }
static int fake_fused(const uint8_t* b, uint64_t nbytes, const fg::DevTables* t, uint32_t strip, int final_, uint64_t* d_offsets, uint64_t cap,
                      uint8_t* scratch, unsigned long long** d_total) {
    unsigned long long* total = reinterpret_cast<unsigned long long*>(scratch);
    *d_total = total;
    total[0] = total[1] = 0;
    if (g_fused_abort_left > 0) {
        --g_fused_abort_left;
        total[1] = 1;
        return 0;
    }
    const uint8_t delim = strip == FG_FRAME_LINE ? 0x0A : 0x00;
    std::vector<uint64_t> offs{0};
    for (uint64_t p = 0; p < nbytes; ++p)
        if (b[p] == delim) offs.push_back(p + 1);
    if (offs.back() != nbytes && final_) offs.push_back(nbytes);
    const uint64_t n = offs.size() - 1;
    total[0] = n;
    const uint64_t rows = n < cap ? n : cap;
    std::vector<uint8_t> bad(rows + 1, 0);
    for (uint64_t i = 0; i < rows; ++i)
        for (uint64_t p = offs[i]; p < offs[i + 1]; ++p) bad[i] |= b[p] >= 0xF8u;
    for (uint64_t i = 0; i <= n && i <= cap + 1; ++i) d_offsets[i] = offs[i];
    fg::DevTables tt = *t;
    return fake_decode(b, offs.data(), rows, &tt, strip, bad.data());
}
extern "C" int fg_launch_rfc5424_fused(const uint8_t* b, uint64_t nbytes, const fg::DevTables* t, const fg::FusedGeom*, hipStream_t, uint64_t*, uint32_t,
                                       uint32_t strip, int final_, uint64_t* d_offsets, uint64_t cap, uint8_t* scratch, const fg_launch_opts*,
                                       unsigned long long** d_total) { return fake_fused(b, nbytes, t, strip, final_, d_offsets, cap, scratch, d_total); }
extern "C" int fg_launch_ltsv_fused(const uint8_t* b, uint64_t nbytes, const fg::DevTables* t, const fg::LtsvDevCfg*, const fg::FusedGeom*, hipStream_t, uint64_t*,
                                    uint32_t, uint32_t strip, int final_, uint64_t* d_offsets, uint64_t cap, uint8_t* scratch, const fg_launch_opts*,
                                    unsigned long long** d_total) { return fake_fused(b, nbytes, t, strip, final_, d_offsets, cap, scratch, d_total); }
extern "C" int fg_launch_gelf_fused(const uint8_t* b, uint64_t nbytes, const fg::DevTables* t, const fg::FusedGeom*, hipStream_t, uint32_t strip, int final_,
                                    uint64_t* d_offsets, uint64_t cap, uint8_t* scratch, const fg_launch_opts*, unsigned long long** d_total) {
    return fake_fused(b, nbytes, t, strip, final_, d_offsets, cap, scratch, d_total);
}
extern "C" int fg_launch_gelf_general_dev(const uint8_t*, const uint64_t*, uint64_t, const fg::DevTables*, hipStream_t, uint32_t, const uint8_t*,
                                          const unsigned long long*) { return 0; }
