// fg_rfc3164.hip -- gfx950 kernel for RFC3164Decoder::decode (rfc3164_decoder.rs:31-213).
//
// The per-line logic is fg_rfc3164_parse.hpp (host + device; checked on the CPU against the oracle).  Launch
// geometry: one 64-lane workgroup = 64 consecutive lines = ONE contiguous byte range of the packed buffer, staged
// into LDS with coalesced 16-byte loads; every lane tokenises ITS line out of LDS (whitespace tokens, month / time
// literals, zone-name lookup in the sorted table, calendar arithmetic) and stores the 68-byte table row (struct of
// arrays: the stores of a wave are coalesced per column).  A group whose bytes exceed the tile (long lines) reads
// from global memory.  RFC3164 produces no structured data: no entry table traffic.
// Roofline: HBM -- line bytes + 8 B offset read once, 68 B written per line.
// (Round 5, measured and dropped: this kernel as a format of the shared streaming pipeline -- persistent waves, the next group
//  prefetched into the register window, chunks by ticket: 3.70 against 3.84 G lines/s at 100 M lines, 47 / 49 / 96 / 357 us against
//  46 / 48 / 101 / 311 us at 16 K .. 1 M lines, profiles/r05ab_rfc3164_pipeline_ab.log.  A group takes ~46 us however it is staged:
//  the time is the lane-serial parser -- three line shapes in one wave, zone names resolved through dependent loads from global
//  memory -- and the one-workgroup form keeps nine to ten such waves on a CU where the pipeline's window leaves seven or eight.)
#include "fg_pipeline.hpp"
#include "fg_rfc3164_parse.hpp"

namespace fg {

// (the regrouped form below)
constexpr uint32_t kSlowKey = 2;         // keys: 0 standard, 1 year first | 2 standard + zone, 3 year first + zone, 4 anything else (the custom form)
constexpr uint32_t kSubLists = 64;       // lists per class of slow lines (the atomics of 2300 resident waves spread over 64 words per class)
// (same box, alternated -- profiles/r06s_rfc3164_small.log: 64 K lines 49 us plain / 71 regrouped, 256 K 101 / 136, 512 K 169 / 171; 16 M 4.17 / 3.13 ms)
constexpr uint64_t kRegroupFrom = 1u << 20;
struct R3164Lists {
    uint32_t* cnt;     // [2 * kSubLists] lines appended to each list (may run past cap: the lines beyond were parsed in place)
    uint4* rec;        // [2 * kSubLists * cap] {line index, offset lo, offset hi, length}
    uint32_t cap;      // records a list holds (a multiple of 64)
};
struct R3164Args {
    r3164::Cfg cfg;
    uint32_t strip;           // FG_FRAME_NONE / _LINE / _NUL: terminators to drop from the frame
    const uint8_t* line_bad;  // [n] 1 = not valid UTF-8 (or null)
};

// the key of a line from its first 32 bytes w[0..7] (zero padded behind its end)
__device__ __forceinline__ uint32_t r3164_key(const uint32_t w[8], uint32_t len) {
    auto at = [&](uint32_t i) -> uint32_t { return i < 32u && i < len ? (w[i >> 2] >> (8u * (i & 3u))) & 0xFFu : 0u; };
    uint32_t q = 0;
    if (at(0) == '<') {  // <PRI>: up to three digits
        q = at(2) == '>' ? 3u : at(3) == '>' ? 4u : at(4) == '>' ? 5u : 0u;
    }
    const uint32_t c0 = at(q), c1 = at(q + 1u), c2 = at(q + 2u), c3 = at(q + 3u);
    const bool upper0 = (c0 - 'A') < 26u, low1 = (c1 - 'a') < 26u, low2 = (c2 - 'a') < 26u;
    uint32_t date_at;
    uint32_t key;
    if (upper0 && low1 && low2 && c3 == ' ') {
        key = 0u;
        date_at = q;
    } else if ((c0 - '0') < 10u && (c1 - '0') < 10u && (c2 - '0') < 10u && (c3 - '0') < 10u && at(q + 4u) == ' ') {
        key = 1u;
        date_at = q + 5u;
    } else {
        return 4u;
    }
    // behind "Mon dd hh:mm:ss " (the day is one or two characters): a token that starts with a capital is likely a zone name
    const uint32_t t1 = date_at + 16u, t0 = date_at + 15u;
    const uint32_t z = at(t0 - 1u) == ' ' ? at(t0) : at(t1);
    return key + ((z - 'A') < 26u ? 2u : 0u);
}

template <class R>
__device__ __forceinline__ void r3164_lane(R rd, uint32_t len, uint64_t li, const DevTables& t, const R3164Args& a) {
    // terminator stripping (BufRead::lines / split(0) semantics)
    if (a.strip != FG_FRAME_NONE && len) {
        const uint32_t b1 = rd.byte(len - 1);
        if (a.strip == FG_FRAME_LINE) {
            if (b1 == '\n') {
                --len;
                if (len && rd.byte(len - 1) == '\r') --len;
            }
        } else if (b1 == 0u) {
            --len;
        }
    }
    r3164::Row r;
    if (a.line_bad && a.line_bad[li]) {
        r.status = FG_ST_BAD_UTF8;
    } else {
        r3164::parse_line(rd, len, a.cfg, r);
    }
    const bool ok = r.status == r3164::ST_OK;
    const fg_span none{0u, FG_NONE};
    t.meta[li] = r.status | (ok ? r.fac : 0xFFu) << 8 | (ok ? r.sev : 0xFFu) << 16 | (ok && r.msg_join ? (uint32_t)FG_F_MSG_JOIN : 0u) << 24;
    t.ts[li] = ok ? r.ts : 0.0;
    t.span[S_HOST][li] = ok ? fg_span{r.host_off, r.host_len} : none;
    t.span[S_APP][li] = none;
    t.span[S_PROC][li] = none;
    t.span[S_MSGID][li] = none;
    t.span[S_MSG][li] = ok ? fg_span{r.msg_off, r.msg_len} : none;
    t.span[S_FULL][li] = ok ? fg_span{0u, r.full_len} : none;
    t.ent_first[li] = 0;
    t.ent_count[li] = 0;
}

// lists.cnt != null (the regrouped form): a lane whose line has a slow shape key hands the line to k_rfc3164_perm instead of parsing it
__global__ __launch_bounds__(kWave) void k_rfc3164(const uint8_t* __restrict__ bytes, const uint64_t* __restrict__ offsets, uint64_t n,
                                                  DevTables t, R3164Args a, uint32_t tile_cap, R3164Lists lists) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_tile[];  // tile_cap + 16 bytes
    const uint64_t g0 = (uint64_t)blockIdx.x * kWave;
    const uint64_t g1 = g0 + kWave < n ? g0 + kWave : n;
    const uint64_t li = g0 + threadIdx.x;
    // the lane's own offsets first (in flight while the tile is staged; fetched after it they are a dependent round trip)
    bool live = li < n;
    const uint64_t lic = live ? li : n - 1u;
    const uint64_t o0 = offsets[lic], o1 = offsets[lic + 1u];
    const uint64_t a_begin = offsets[g0], a_end = offsets[g1];
    const uint64_t a0 = a_begin & ~15ull;
    const uint64_t span = (a_end - a0 + 15ull) & ~15ull;
    if (span <= (uint64_t)tile_cap) {
        stage_tile<20>(bytes, a0, (uint32_t)span, s_tile);  // (20 KiB per round trip: the usual tile in one)
        __syncthreads();  // single-wave workgroup: orders the LDS writes before the lanes' reads
        LdsReader rd(reinterpret_cast<const uint32_t*>(s_tile), (uint32_t)(o0 - a0));
        if (lists.cnt) {  // wave-uniform
            const uint32_t lane = threadIdx.x;
            const uint64_t len64 = o1 - o0;
            uint32_t key = 0u;
            if (live) {
                uint32_t w[8];
                rd.load16(0u, w);
                rd.load16(16u, w + 4);  // (what lies behind a line's end is masked by its length; the tile is followed by 48 spare bytes)
                key = r3164_key(w, len64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)len64);
            }
            // a group that is MOSTLY slow shapes parses them where they are: the wave is full enough as it stands, and a corpus of one slow
            // shape would otherwise go through the lists line by line for nothing (16.7 M zone-tagged lines: 7.2 G lines/s in place, 4.9 G
            // through the lists; the custom form 5.3 vs 3.9 -- profiles/r06fin_rfc3164_shapes.log)
            const bool regroup = __popcll(__ballot(live && key >= kSlowKey)) < 32;  // wave-uniform
#pragma unroll
            for (uint32_t c = 0; c < 2u; ++c) {
                const bool mine = regroup && live && len64 <= 0xFFFFFFFFull && (c == 0u ? (key == 2u || key == 3u) : key == 4u);
                const unsigned long long m = __ballot(mine);
                if (m) {  // wave-uniform
                    const uint32_t s = c * kSubLists + (blockIdx.x & (kSubLists - 1u));
                    uint32_t base = 0u;
                    if (lane == 0u) base = atomicAdd(lists.cnt + s, (uint32_t)__popcll(m));
                    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                    const uint32_t slot = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                    if (mine && slot < lists.cap) {  // (a full list: the line is parsed here after all)
                        lists.rec[(uint64_t)s * lists.cap + slot] = make_uint4((uint32_t)li, (uint32_t)o0, (uint32_t)(o0 >> 32), (uint32_t)len64);
                        live = false;
                    }
                }
            }
        }
        if (live) r3164_lane(rd, (uint32_t)(o1 - o0), li, t, a);
    } else if (live) {
        GlobalReader rd(reinterpret_cast<const uint32_t*>(bytes), o0);
        r3164_lane(rd, (uint32_t)(o1 - o0), li, t, a);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The SLOW SHAPES REGROUPED (round 6).  A wave runs the union of the shapes its 64 lanes hold: batches of ONE shape each
// decode at 11.6 G lines/s (`<pri>Mon dd hh:mm:ss host app: msg`, 68 % of the bench corpus), 7.2 G (the same with a zone name behind the
// time, 12 %) and 5.2 G (the custom form, 20 %: the standard parse fails first, then the ": " search and a second date parse) -- and their
// MIX at 3.8-4.1 G, below any of them (tools/probe/rfc3164_shapes.py at 16 M lines; at 1 M lines the corpus sits in the 256 MB memory-side
// cache and everything reads ~20 % lower).  The shape shows in the first bytes of a line (r3164_key: a scheduling hint, nothing more -- the
// parser is the same exact parser whatever the key says), and the kernel has those bytes in LDS anyway:
//   k_rfc3164       (lists != null) a lane whose line has a SLOW key does not parse it: the wave appends the line -- {index, offset, length}
//                   -- to one of 64 lists of its class (zone-ish / custom; one atomic per wave and class, on 64 words per class: a few
//                   per microsecond each) and the everyday lines of the wave are parsed at the everyday line's pace
//   k_rfc3164_perm  the listed lines, 64 of ONE class to the workgroup: each line staged as a row of its own in LDS (sixteen lanes to the
//                   line, every load of the group in flight at once), the unchanged per-lane parser, rows to the lines' own positions.
// Steps to here, each measured on one box against the plain kernel's 3.82 G lines/s (profiles/r06m..q_rfc3164*.log, 16 M lines): every line
// through a key-sorted permutation built by a pre-pass (classify + scan + scatter) 3.22 G staged row by row, 4.1-4.2 G with all loads of a
// group in flight (the permuted decode is ~40 % slower per line than the contiguous one, and the pre-pass reads 58 % of the stream again);
// the everyday lines in place and only the slow ones permuted 4.45 G; then the key computed IN the decode kernel from the staged tile --
// no pre-pass at all -- (below).  Round 5's two forms for comparison: the streaming pipeline -4 %, the two-kernel form that parsed every
// line the standard way first +5 %.  MEASURED AND REMOVED at the end of round 6 (profiles/r06au_rfc3164_shapes.log, "library choice" there):
// the regrouping done INSIDE a workgroup of four waves (four tiles, the 256 lines dealt out again in key order through LDS, no lists, no
// second kernel) -- 3.49 G lines/s on the mix against 4.10 plain and 5.18 with the lists, and 8.8 against 11.5 G on the everyday shape
// alone: the workgroup holds its four tiles until its slowest (custom-shape) wave is done, and two workgroups per CU overlap their
// staging and their parsing far worse than nine independent waves do.  Small batches (below kRegroupFrom lines: one M) keep the plain kernel -- the second launch and its part-filled workgroups cost more than the union does.
// ---------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWave) void k_rfc3164_perm(const uint8_t* __restrict__ bytes, DevTables t, R3164Args a, uint32_t tile_cap, R3164Lists lists) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_tile[];  // tile_cap + 16 bytes
    const uint32_t lane = threadIdx.x;
    // workgroup -> (list, 64 records of it): how many a list holds is known on the device only -- the grid covers every list's capacity
    // and the workgroups behind a list's last record leave at once
    const uint32_t per_list = lists.cap / kWave;
    const uint32_t s = blockIdx.x / per_list, j = blockIdx.x - s * per_list;
    uint32_t cnt = lists.cnt[s];
    if (cnt > lists.cap) cnt = lists.cap;
    const uint32_t g0 = j * kWave;
    if (g0 >= cnt) return;
    const uint32_t n = cnt;
    const bool live = g0 + lane < n;
    const uint4 pr = lists.rec[(uint64_t)s * lists.cap + (live ? g0 + lane : n - 1u)];
    const uint64_t li = pr.x;
    const uint64_t o0 = (uint64_t)pr.y | ((uint64_t)pr.z << 32), o1 = o0 + pr.w;
    // the line's row in the tile: from its first 16-byte boundary, its length rounded up to 16; rows packed by a wave prefix sum
    const uint32_t al = (uint32_t)(o0 & 15ull);
    const uint64_t want = (o1 - o0) + al;
    uint32_t st = live ? (want > 0x7FFFFFF0ull ? 0x7FFFFFF0u : (uint32_t)((want + 15ull) & ~15ull)) : 0u;
    if (st == 0u && live) st = 16u;  // (an empty line still gets a row: the reader may look at its first dword)
    uint32_t total;
    const uint32_t tb = wv::excl_sum(st, &total);
    const bool in_tile = live && tb + st <= tile_cap;  // (rows are laid out in lane order: a row that does not fit, and those behind it, read global memory)
    // stage: SIXTEEN lanes to the line, sixteen bytes to the lane -- a wave-instruction moves 256 bytes of four lines, all sixteen
    // instructions of a pass (the 64 lines' first 256 bytes of row) are in flight before the first store: ONE round trip for the
    // everyday group.  (The first cut staged row by row, four rows per round trip: sixteen dependent trips to memory per workgroup,
    // and the kernel ran no faster than the mixed one -- profiles/r06m_rfc3164_regroup.log.)  Longer rows take further passes.
    uint4* dst = reinterpret_cast<uint4*>(s_tile);
    const uint32_t sub = lane >> 4, c16 = (lane & 15u) * 16u;
    const uint32_t o0_lo = (uint32_t)o0, o0_hi = (uint32_t)(o0 >> 32);
    const uint32_t st_fit = in_tile ? st : 0u;  // (a row that does not fit is not staged)
    // rows longer than 256 bytes (a third of a 128 .. 320-byte corpus: a second pass over ALL rows for them cost every workgroup a second
    // round trip): their second 256 bytes ride in the SAME round trip, four such rows per wave-instruction, up to thirty-two of them
    const unsigned long long longm = __ballot(st_fit > 256u);
    constexpr uint32_t kLong = 6;  // wave-instructions of second halves in the first round trip: 24 long rows (a third of 64 is 21)
    // what a helper lane needs of the line it stages -- where it lies, where its row goes -- through LDS: ONE 16-byte read per step
    // (three ds_bpermute each before), and the list of the long rows (lane ids in ballot order)
    __shared__ uint4 s_meta[kWave];
    __shared__ uint8_t s_long[kWave];
    s_meta[lane] = make_uint4(o0_lo, o0_hi, tb | (st_fit << 16), 0u);
    if (st_fit > 256u) s_long[__popcll(longm & ((1ull << lane) - 1ull))] = (uint8_t)lane;
    __syncthreads();
    uint4 v[16], v2[kLong];
    uint32_t rpk[16], r2pk[kLong];  // row offset | row bytes << 16 (the tile is below 64 KiB)
#pragma unroll
    for (uint32_t i = 0; i < 16u; ++i) {
        const uint4 m = s_meta[i * 4u + sub];  // the line this lane helps to stage in step i
        rpk[i] = m.z;
        const uint64_t b = (((uint64_t)m.x | ((uint64_t)m.y << 32)) & ~15ull) + c16;
        v[i] = make_uint4(0u, 0u, 0u, 0u);
        if (c16 < (rpk[i] >> 16)) v[i] = *reinterpret_cast<const uint4*>(bytes + b);
    }
    const uint32_t n_long = (uint32_t)__popcll(longm);
#pragma unroll
    for (uint32_t i = 0; i < kLong; ++i) {
        r2pk[i] = 0u;
        v2[i] = make_uint4(0u, 0u, 0u, 0u);
        if (i * 4u < n_long) {  // wave-uniform
            const uint32_t k = i * 4u + sub;
            const bool has = k < n_long;
            const uint4 m = s_meta[has ? (uint32_t)s_long[k] : 0u];
            r2pk[i] = has ? m.z : 0u;
            const uint64_t b = (((uint64_t)m.x | ((uint64_t)m.y << 32)) & ~15ull) + 256u + c16;
            if (256u + c16 < (r2pk[i] >> 16)) v2[i] = *reinterpret_cast<const uint4*>(bytes + b);
        }
    }
#pragma unroll
    for (uint32_t i = 0; i < 16u; ++i)
        if (c16 < (rpk[i] >> 16)) dst[((rpk[i] & 0xFFFFu) + c16) >> 4] = v[i];
#pragma unroll
    for (uint32_t i = 0; i < kLong; ++i)
        if (256u + c16 < (r2pk[i] >> 16)) dst[((r2pk[i] & 0xFFFFu) + 256u + c16) >> 4] = v2[i];
    // what is left: rows beyond 512 bytes, and long rows beyond the twenty-fourth -- pass by pass over all rows (rare)
    uint32_t rest = st_fit > 512u || (st_fit > 256u && (uint32_t)__popcll(longm & ((1ull << lane) - 1ull)) >= 4u * kLong) ? st_fit : 0u;
    uint32_t max_rest = rest;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)max_rest, d, kWave);
        max_rest = o > max_rest ? o : max_rest;
    }
    for (uint32_t base = 256u; base < max_rest; base += 256u) {  // wave-uniform
#pragma unroll
        for (uint32_t i0 = 0; i0 < 16u; i0 += 4u) {
            uint4 w[4];
            uint32_t wtb[4], wst[4];
#pragma unroll
            for (uint32_t j = 0; j < 4u; ++j) {
                const uint32_t r = (i0 + j) * 4u + sub;
                const uint32_t lo = (uint32_t)__shfl((int)o0_lo, (int)r, kWave), hi = (uint32_t)__shfl((int)o0_hi, (int)r, kWave);
                wtb[j] = (uint32_t)__shfl((int)tb, (int)r, kWave);
                wst[j] = (uint32_t)__shfl((int)rest, (int)r, kWave);
                const uint64_t b = (((uint64_t)lo | ((uint64_t)hi << 32)) & ~15ull) + base + c16;
                w[j] = make_uint4(0u, 0u, 0u, 0u);
                if (base + c16 < wst[j]) w[j] = *reinterpret_cast<const uint4*>(bytes + b);
            }
#pragma unroll
            for (uint32_t j = 0; j < 4u; ++j)
                if (base + c16 < wst[j]) dst[(wtb[j] + base + c16) >> 4] = w[j];
        }
    }
    __syncthreads();
    if (live) {
        if (in_tile) {
            LdsReader rd(reinterpret_cast<const uint32_t*>(s_tile), tb + al);
            r3164_lane(rd, (uint32_t)(o1 - o0), li, t, a);
        } else {
            GlobalReader rd(reinterpret_cast<const uint32_t*>(bytes), o0);
            r3164_lane(rd, (uint32_t)(o1 - o0), li, t, a);
        }
    }
}

}  // namespace fg

static inline uint32_t r3164_list_cap(uint64_t n) {  // half the lines per class, spread over the class's lists, in whole workgroups
    const uint64_t c = (n / 2u + fg::kSubLists - 1u) / fg::kSubLists;
    const uint64_t r = (c + fg::kWave - 1u) / fg::kWave * fg::kWave;
    return (uint32_t)(r < fg::kWave ? fg::kWave : r);
}
extern "C" uint64_t fg_rfc3164_regroup_from(void) { return fg::kRegroupFrom; }
extern "C" uint64_t fg_rfc3164_scratch_bytes(uint64_t n) { return 1024u + (uint64_t)2u * fg::kSubLists * r3164_list_cap(n) * 16u; }
// scratch (fg_rfc3164_scratch_bytes(n) bytes, or null) and regroup: 0 = the library's choice (from kRegroupFrom lines on), 1 = always
// (tests), 2 = never (A/B)
extern "C" int fg_launch_rfc3164(const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t n, const fg::DevTables* t,
                                 const fg::r3164::Cfg* cfg, uint32_t tile_cap, hipStream_t stream, uint32_t strip,
                                 const uint8_t* line_bad, uint8_t* scratch, int regroup) {
    if (n == 0) return 0;
    const uint64_t blocks = (n + fg::kWave - 1) / fg::kWave;
    if (blocks > 0x7FFFFFFFull) return -1;
    fg::R3164Args a{*cfg, strip, line_bad};
    const bool grouped = scratch && regroup != 2 && n < 0xFFFFFFFFull && (regroup == 1 || n >= fg::kRegroupFrom);
    fg::R3164Lists lists{nullptr, nullptr, 0u};
    if (grouped) {
        lists.cnt = reinterpret_cast<uint32_t*>(scratch);
        lists.rec = reinterpret_cast<uint4*>(scratch + 1024u);
        lists.cap = r3164_list_cap(n);
        if (hipMemsetAsync(scratch, 0, 1024u, stream) != hipSuccess) return -1;
    }
    hipLaunchKernelGGL(fg::k_rfc3164, dim3((uint32_t)blocks), dim3(fg::kWave), tile_cap + 48u, stream, d_bytes, d_offsets, n, *t, a, tile_cap, lists);
    if (grouped) {
        // the slow shapes, 64 of a class to the workgroup
        const uint64_t pblocks = (uint64_t)2u * fg::kSubLists * (lists.cap / fg::kWave);
        hipLaunchKernelGGL(fg::k_rfc3164_perm, dim3((uint32_t)pblocks), dim3(fg::kWave), tile_cap + 48u, stream, d_bytes, *t, a, tile_cap, lists);
    }
    return (int)hipGetLastError();
}
