// fg_fused.hpp -- the streaming skeleton of the decode kernels that FRAME THE STREAM THEMSELVES (round 6): persistent_loop's sibling.
//
// persistent_loop (fg_pipeline.hpp) takes lines from an offsets array that a framing pass wrote first: the stream is read twice on the
// device and the frame count visits the host between the two kernels.  Here a wave takes a byte TILE of the raw
// stream (fg_fuse.hpp: the tile, 16 bytes in front, `look` bytes behind), and stage A -- the same register window, the same LDS
// tile, the same per-format class bitmaps -- also leaves one delimiter | UTF-8-error word per 16-byte chunk.  The lines that START in
// the tile come out of a count + list pass over those words; stage B is the format's unchanged decode() over a GroupCtx that looks
// exactly like persistent_loop's (frames with their terminators, stripped as fg_decode_frames_device strips them); a tile with more
// lines than a pass takes runs stage B again over the same staged tile.
//
// Row indices.  Row i of the tables is line i of the stream, so a tile must know how many lines start before it:
//   * the tile's own count is published right after its stage A:  tcount[T] = count + 1  and one atomic add of
//     (1 << 40 | count) to the aggregate of its BLOCK of 64 tiles;
//   * the look-back is two-level and both levels are one step in the steady state: the counts of the tiles before T in its block
//     (64 lanes read 64 words), and the lines before the block = the aggregates of the complete blocks back to the nearest block whose
//     inclusive prefix is published (64 lanes read 64 blocks = 4096 tiles, more than a grid of persistent waves holds in flight).
//     (One level -- the one-pass framing scan's chain, fg_frame.hip -- moves 64 tile descriptors per step: with ~1800 tiles of 15 KiB
//     in flight at once, all counted and none prefixed, a tile would walk a dozen steps of ~1 us each per 5 us of streaming.)
//   * it is done one iteration LATE, after the next tile's stage A and before the rows of the tile are stored: every tile before has
//     published its count long before, nothing of this wave is in flight at that point, and the wave never spins in practice.
// Tiles are DRAWN from K ticket counters (wave w draws from counter w mod K; counter q hands out tiles q, q + K, q + 2K ...), all of
// them, the first included: a tile is only ever held by a running wave, tiles go out in order per counter, and the same-address atomic
// rate of one word (a few dozen per microsecond) is never approached.  Forward progress: the smallest unpublished tile is
// either held by a running wave -- which publishes after its stage A without waiting for anybody -- or undrawn, and then a running wave of
// its counter (every residue mod K is among the first K workgroups) holds a smaller, published tile and draws it next.  The wait is
// BOUNDED all the same: a wave that polls kSpinLimit times raises the abort word, every wave leaves, and the host path falls back to
// the separate framing + decode kernels.
#pragma once
#include "fg_fuse.hpp"
#include "fg_pipeline.hpp"

namespace fg {

constexpr uint32_t kFusedSpinLimit = 1u << 18;     // polls (~0.2 s) before a waiting wave gives up
constexpr unsigned long long kAggCount = (1ull << 40) - 1ull;  // low bits of a block aggregate: lines; above: tiles published
constexpr unsigned long long kPreFlag = 1ull << 63;

struct FusedArgs {
    uint64_t nbytes;            // bytes of the stream chunk
    uint64_t ntiles;            // ceil(nbytes / S)
    uint32_t S, look;           // tile geometry (fuse::plan_tile)
    uint32_t delim4;            // the terminator byte in all four bytes of a dword
    uint32_t final_;            // nonzero: the stream ends with this chunk (an unterminated last piece is a frame)
    uint32_t lds_off;           // dynamic-LDS offset of the fused block (fuse::carve)
    uint32_t ext;               // bytes staged on at a time behind the look-ahead (FusedGeom::ext)
    uint32_t counters;          // K (<= kFusedCounters, <= the grid)
    uint64_t cap;               // rows the tables hold; offsets holds cap + 2
    uint64_t* offsets;          // out: offsets[i] = start of frame i, offsets[total] = end of the last frame
    uint32_t* tcount;           // [ntiles]   0 = not yet, else the tile's line count + 1
    unsigned long long* bagg;   // [nblocks]  tiles published << 40 | their lines
    unsigned long long* bpre;   // [nblocks]  kPreFlag | lines up to the END of the block
    uint32_t* tickets;          // [K * kFusedCounterStride]
    unsigned long long* total;  // [0] lines of the stream  [1] nonzero: a wave gave up waiting (nothing is valid)
};

inline void fused_carve(uint8_t* scratch, uint64_t nbytes, uint32_t S, FusedArgs* fa) {
    const uint64_t nt = fused_tiles(nbytes, S), nb = (nt + 63u) / 64u;
    fa->tickets = reinterpret_cast<uint32_t*>(scratch);
    fa->total = reinterpret_cast<unsigned long long*>(scratch + kFusedCounters * kFusedCounterStride * 4u);
    fa->tcount = reinterpret_cast<uint32_t*>(scratch + kFusedCounters * kFusedCounterStride * 4u + 128u);  // (total: 16 words -- the two results and the measurement variant's)
    fa->bagg = reinterpret_cast<unsigned long long*>(scratch + kFusedCounters * kFusedCounterStride * 4u + 128u + ((nt * 4u + 63u) & ~63ull));
    fa->bpre = fa->bagg + nb;
    fa->ntiles = nt;
}

// -DFG_FUSED_STATS (a measurement variant, FG_BUILD_VARIANT=stats; never the product): event counts and shader-clock phases of the fused loop,
// accumulated per wave and added to total[2 ..] at the wave's end: [2] tiles [3] slow look-backs [4] tiles that staged on [5] tail scans
// [6] passes of stage B [7] tiles with a UTF-8 error bit; cycles: [8] wait for the window [9] stage A [10] count + tail [11] prefetch issue +
// publish [12] look-back + row stores [13] list + stage B [14] slow look-backs because a tile of the block had not published [15] ... because no prefix / an incomplete block lay in the 64 blocks
#if defined(FG_FUSED_STATS)
#define FG_ST(k, v) (st_acc[k] += (v))
#define FG_CLK(k)                                              \
    do {                                                       \
        const uint64_t now_ = __builtin_amdgcn_s_memtime();    \
        st_acc[k] += now_ - st_clk;                            \
        st_clk = now_;                                         \
    } while (0)
#else
#define FG_ST(k, v) ((void)0)
#define FG_CLK(k) ((void)0)
#endif

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
__device__ __forceinline__ uint64_t wave_sum64(uint64_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, kWave);
    return v;
}

// ---- rare paths, OUT OF LINE (everything by value: a kernel-argument struct whose address reaches a function the compiler does not
//      inline lives in scratch memory for the whole kernel) ----
// the tile's last line runs past everything the tile could stage: where it ends (fuse::forward_scan through global memory)
struct TailEnd { uint64_t end; uint32_t bad; };
__device__ __noinline__ TailEnd fused_tail_scan(const uint8_t* bytes, uint64_t padded, uint64_t nbytes, uint32_t delim4, uint64_t p0, uint32_t pw0) {
    auto ld = [&](uint64_t pos) -> fuse::U4 {
        fuse::U4 q{0u, 0u, 0u, 0u};
        if (pos + 16u <= padded) {
            const uint4 x = *reinterpret_cast<const uint4*>(bytes + pos);
            q = fuse::U4{x.x, x.y, x.z, x.w};
        }
        return q;
    };
    bool bad = false;
    const uint64_t e = fuse::forward_scan(ld, p0, nbytes, delim4, pw0, &bad);
    return TailEnd{e, bad ? 1u : 0u};
}
// a UTF-8 error inside tile positions [a, b)?  (only a tile in which stage A saw an error bit asks)
__device__ __noinline__ bool fused_line_bad(const uint32_t* tile32, uint64_t g_base, uint32_t g_span, uint32_t g_own_end, uint32_t g_end_x, uint32_t a, uint32_t b) {
    const fuse::Geo g{g_base, g_span, g_own_end, g_end_x};
    return fuse::line_bad(tile32, g, a, b);
}

// The look-back of tile T, in two halves so that its loads ride behind loads the wave waits for anyway:
//   lookback_issue   asks for the counts of the tiles before T in its block and for the aggregates / prefixes of the 64 blocks before
//   lookback_finish  lines of the stream that start before tile T (wave-uniform); publishes what the tiles behind can use; polls on
//                    (bounded) in the rare case that a tile before has not published yet.  *aborted: gave up.
struct LookBack {
    uint32_t v;                   // tcount of tile (block start + lane), 1 for lanes >= j
    unsigned long long agg, pre;  // of block B - 1 - lane
};
__device__ __forceinline__ LookBack lookback_issue(const FusedArgs& fa, uint64_t T) {
    const uint32_t lane = threadIdx.x;
    const uint64_t B = T >> 6;
    const uint32_t j = (uint32_t)(T & 63u);
    LookBack lb;
    lb.v = 1u;
    if (lane < j) lb.v = __hip_atomic_load(fa.tcount + (B << 6) + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int64_t b = (int64_t)B - 1 - (int64_t)lane;
    lb.agg = 64ull << 40, lb.pre = kPreFlag;  // (before the stream: a complete block of no lines with prefix 0)
    if (b >= 0) {
        lb.agg = __hip_atomic_load(fa.bagg + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        lb.pre = __hip_atomic_load(fa.bpre + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return lb;
}
// the whole look-back with polling, by value and out of line: the slow path of lookback_finish and the look-back of a tile that takes
// several passes.  Returns ~0ull when it gave up (the abort word is raised).
__device__ __noinline__ uint64_t lookback_slow(uint32_t* tcount, unsigned long long* bagg, unsigned long long* bpre, unsigned long long* total,
                                               uint64_t ntiles, uint64_t T, uint32_t n_own) {
    const uint32_t lane = threadIdx.x;
    const uint64_t B = T >> 6;
    const uint32_t j = (uint32_t)(T & 63u);
    uint32_t polls = 0;
    auto give_up = [&]() -> bool {  // one more poll; true = leave
        ++polls;
        if (polls > kFusedSpinLimit) {
            if (lane == 0u) __hip_atomic_store(total + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return true;
        }
        if ((polls & 15u) == 0u && __hip_atomic_load(total + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull) return true;
        __builtin_amdgcn_s_sleep(2);
        return false;
    };
    // ---- the tiles before T in its block ----
    uint32_t within = 0;
    for (;;) {
        uint32_t v = 1u;
        if (lane < j) v = __hip_atomic_load(tcount + (B << 6) + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__ballot(v == 0u) == 0ull) {
            uint32_t tot;
            (void)wv::excl_sum(v - 1u, &tot);
            within = tot;
            break;
        }
        if (give_up()) return ~0ull;
    }
    // ---- the blocks before B: complete aggregates back to the nearest published prefix ----
    uint64_t pre = 0;
    int64_t idx = (int64_t)B - 1;
    while (idx >= 0) {
        const int64_t b = idx - (int64_t)lane;
        unsigned long long agg = 64ull << 40, p = kPreFlag;  // (before the stream: a complete block of no lines with prefix 0)
        if (b >= 0) {
            agg = __hip_atomic_load(bagg + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            p = __hip_atomic_load(bpre + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const bool have_p = (p & kPreFlag) != 0ull;
        const unsigned long long mp = __ballot(have_p);
        const uint32_t first = mp ? (uint32_t)__builtin_ctzll(mp) : 64u;  // the nearest block that knows its prefix
        const bool part = lane < first;                                    // blocks between here and there: their aggregates
        if (__ballot(part && (agg >> 40) != 64ull) != 0ull) {             // ... one of them is still being counted
            if (give_up()) return ~0ull;
            continue;
        }
        pre += wave_sum64(part ? (uint64_t)(agg & kAggCount) : 0ull);
        if (mp) {
            const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)p, (int)first, kWave), hi = (uint32_t)__shfl((int)(uint32_t)(p >> 32), (int)first, kWave);
            pre += (((uint64_t)hi << 32) | lo) & ~kPreFlag;
            break;
        }
        idx -= kWave;
    }
    const uint64_t base = pre + within;
    if (lane == 0u) {
        const uint64_t in_block = ntiles - (B << 6) < 64u ? ntiles - (B << 6) : 64u;
        if (j + 1u == in_block) __hip_atomic_store(bpre + B, kPreFlag | (base + n_own), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (j == 0u && B != 0u) __hip_atomic_store(bpre + (B - 1u), kPreFlag | pre, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (T + 1u == ntiles) __hip_atomic_store(total, (unsigned long long)(base + n_own), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return base;
}
// the steady state in line: every tile before has published and a prefix lies within the 64 blocks asked for -- one pass over what
// lookback_issue brought; anything else goes to lookback_slow
__device__ __forceinline__ uint64_t lookback_finish(const FusedArgs& fa, uint64_t T, uint32_t n_own, const LookBack& lb, bool* aborted) {
    const uint32_t lane = threadIdx.x;
    const uint64_t B = T >> 6;
    const uint32_t j = (uint32_t)(T & 63u);
    const bool have_p = (lb.pre & kPreFlag) != 0ull;
    const unsigned long long mp = __ballot(have_p);
    const uint32_t first = mp ? (uint32_t)__builtin_ctzll(mp) : 64u;
    const bool part = lane < first;
    const bool ready = __ballot(lb.v == 0u) == 0ull && mp != 0ull && __ballot(part && (lb.agg >> 40) != 64ull) == 0ull;  // wave-uniform
    if (!ready) {
#if defined(FG_FUSED_STATS)
        if (lane == 0u) {
            atomicAdd(fa.total + 3, 1ull);
            if (__ballot(lb.v == 0u) != 0ull) atomicAdd(fa.total + 14, 1ull);
            else atomicAdd(fa.total + 15, 1ull);
        }
#endif
        const uint64_t r = lookback_slow(fa.tcount, fa.bagg, fa.bpre, fa.total, fa.ntiles, T, n_own);
        if (r == ~0ull) *aborted = true;
        return r;
    }
    uint32_t within;
    (void)wv::excl_sum(lb.v - 1u, &within);
    uint64_t pre = wave_sum64(part ? (uint64_t)(lb.agg & kAggCount) : 0ull);
    const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)lb.pre, (int)first, kWave), hi = (uint32_t)__shfl((int)(uint32_t)(lb.pre >> 32), (int)first, kWave);
    pre += (((uint64_t)hi << 32) | lo) & ~kPreFlag;
    const uint64_t base = pre + within;
    if (lane == 0u) {
        const uint64_t in_block = fa.ntiles - (B << 6) < 64u ? fa.ntiles - (B << 6) : 64u;
        // the block's last tile publishes the block's prefix; its first one the prefix of the block before (if nobody has)
        if (j + 1u == in_block) __hip_atomic_store(fa.bpre + B, kPreFlag | (base + n_own), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (j == 0u && B != 0u) __hip_atomic_store(fa.bpre + (B - 1u), kPreFlag | pre, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (T + 1u == fa.ntiles) __hip_atomic_store(fa.total, (unsigned long long)(base + n_own), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return base;
}
__device__ __forceinline__ uint64_t fused_lookback(const FusedArgs& fa, uint64_t T, uint32_t n_own, bool* aborted) {
    const uint64_t r = lookback_slow(fa.tcount, fa.bagg, fa.bpre, fa.total, fa.ntiles, T, n_own);
    if (r == ~0ull) *aborted = true;
    return r;
}

// F as for persistent_loop.  `strip` = the framing (FG_FRAME_LINE / FG_FRAME_NUL).  L = lines a pass of stage B takes (<= 64).
template <int NB, class F>
__device__ __forceinline__ void fused_loop(const uint8_t* __restrict__ bytes, const DevTables& t, uint32_t tile_cap, uint32_t L, F& fmt,
                                           const FusedArgs& fa, uint32_t strip, uint64_t* stash_base) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint16_t* bm16 = reinterpret_cast<uint16_t*>(smem + tile_cap + 64u);
    const uint32_t bm_stride = tile_cap / 16u + 16u;
    uint4* dst = reinterpret_cast<uint4*>(smem);
    const uint32_t lane = threadIdx.x;
    // (the RFC5424 decoder keeps structured-data records in the tile itself and only asks WHETHER it has a stash)
    uint64_t* const stash = stash_base ? stash_base + (uint64_t)blockIdx.x * (kStashEntries * kStashWords * kWave) : nullptr;
    const fuse::Lds FL = fuse::carve(smem + fa.lds_off, tile_cap);
    const uint32_t dm_cap = 64u * FL.R;
    uint32_t* ent_state = reinterpret_cast<uint32_t*>(smem + tile_cap + 56u);
    if (lane < 2u) ent_state[lane] = 0u;
    for (uint32_t c = lane; c < dm_cap; c += kWave) FL.dm16[c] = 0u;
    const uint32_t K = fa.counters;
    // (workgroups go round the eight XCDs: with `blockIdx.x % K` a counter's waves all sat on ONE XCD, the XCDs do not run at one pace,
    //  and every tile waited for the slowest counter's; `(blockIdx.x / 8) % K` gives each counter an eighth of every XCD's waves.  A grid
    //  of fewer than 8 K workgroups keeps the plain residue: every counter must have a wave)
    const uint32_t cq = K ? (gridDim.x >= 8u * K ? (blockIdx.x >> 3) % K : blockIdx.x % K) : 0u;
    uint32_t* const counter = fa.tickets + cq * kFusedCounterStride;
    const uint32_t term4 = fa.delim4;
    const uint64_t padded = (fa.nbytes + 15ull) & ~15ull;

    // K == 0 (FG_LO_STATIC_CHUNKS; A/B): tiles dealt out round-robin, no tickets -- every wave of the grid must then be resident for the
    // look-back to make progress (it is, on an otherwise idle device; the bounded wait covers the rest)
    const bool dyn = K != 0u;
    uint32_t tk_raw = 0;
    uint64_t static_next = blockIdx.x;
    auto draw = [&]() {
        if (dyn) {
            if (lane == 0u) tk_raw = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    auto drawn = [&]() -> uint64_t {
        if (dyn) return (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)tk_raw) * K + cq;
        const uint64_t r = static_next;
        static_next += gridDim.x;
        return r;
    };
    // the register window of tile T: rows of 1 KiB from the 16 bytes in front of the tile on, bounded by the readable range of the stream
    auto load_window = [&](uint64_t T, const fuse::Geo& g, u32x4* v) {
        // (window chunk c = staged chunk c + sh: the stream's first tile has no 16 bytes in front of it -- its window starts AT the stream
        //  and stage A shifts it by one chunk behind a synthetic one; every row's offset stays lane * 16 + an immediate)
        const uint32_t pre = T ? 0u : fuse::kPre;
        const uint64_t p0 = T ? g.base : 0ull;  // stream position of the first byte fetched
        const uint64_t readable = padded - p0;
        const uint32_t want = g.span - pre;
        const uint32_t range = readable < (uint64_t)want ? (uint32_t)readable : want;
        __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(bytes + p0), (short)0, (int)range, 0x00020000);
#pragma unroll
        for (int k = 0; k < NB; ++k) v[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(lane * 16u + k * 1024u), 0, FG_STREAM_AUX);
    };

#if defined(FG_FUSED_STATS)
    uint64_t st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t st_clk = __builtin_amdgcn_s_memtime();
#endif
    draw();
    uint64_t T = drawn();
    if (T >= fa.ntiles) return;
    draw();  // (the tile after: asked for now, looked at after this tile's stage A)
    fuse::Geo g = fuse::tile_geo(T, fa.S, fa.look, fa.nbytes);
    u32x4 v[NB];
    load_window(T, g, v);

    // what waits for its row index: the rows of the LAST pass of the tile before (stored after this tile's stage A)
    RowOut pend{};
    uint64_t pend_o0 = 0, pend_o1 = 0;
    uint32_t pend_k = 0;       // the lane's line index inside its tile
    bool pend_valid = false;   // this lane holds a row
    bool pend_any = false;     // a tile waits for its look-back (wave-uniform)
    uint64_t pend_T = 0;
    uint32_t pend_nown = 0;
    uint32_t prev_nchunk = 0;
    bool aborted = false;

    auto store_rows = [&](uint64_t base, const RowOut& o, bool valid, uint32_t k, uint64_t o0, uint64_t o1) {
        const uint64_t li = base + k;
        if (valid && li < fa.cap) {
            store_row(t, li, o);
            fa.offsets[li] = o0;
        }
        // the end of the last frame of the pass: the next frame's start, or -- behind the stream's last frame -- where the frames end
        const unsigned long long m = __ballot(valid);
        if (m && lane == 63u - (uint32_t)__builtin_clzll(m) && li + 1u <= fa.cap + 1u) fa.offsets[li + 1u] = o1;
    };

    for (;;) {
        // ---- the look-back of the tile before: its loads ride behind this tile's window, which the wave waits for anyway ----
        FG_MARK(T);
        FG_CLK(13);
        LookBack lbk{1u, 0ull, 0ull};
        if (pend_any) lbk = lookback_issue(fa, pend_T);
#if defined(FG_FUSED_STATS)
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the window has landed
        FG_CLK(8);
#endif
        FG_ST(2, 1);
        // ---- stage A: registers -> LDS, the format's classes and the delimiter mask per chunk ----
        const uint32_t nchunk = g.span >> 4;
        const uint32_t sh = T ? 0u : 1u;                                  // staged chunk = window chunk + sh
        const uint32_t nrow = (nchunk - sh + kWave - 1u) / kWave;         // rows of the window that hold staged chunks
        uint32_t row_last = 0u;  // the last dword of the row before (the UTF-8 rules look three bytes back)
        uint32_t err_acc = 0u;   // UTF-8 error bits this lane saw
        // (the hot path knows nothing of the stream's end: the one tile that holds it is put right below, out of LDS)
        auto stage = [&](const uint4& q, uint32_t idx) {
            dst[idx] = q;
            F::classify_store(q, bm16, idx, bm_stride, idx < nchunk ? term4 : wv::kPastSpan);
            const uint32_t pw = wv::shfl_up1(q.w, row_last);
            row_last = (uint32_t)__builtin_amdgcn_readlane((int)q.w, 63);
            uint32_t m = fuse::chunk_masks(q.x, q.y, q.z, q.w, pw, term4, 16);
            if (idx == 0u) m = fuse::pre_chunk_mask(m, false);
            if (idx >= nchunk) m = 0u;
            err_acc |= m;
            if (idx < dm_cap) FL.dm16[idx] = (uint16_t)m;
        };
        if (sh && lane == 0u) {  // the stream's first tile: nothing lies before it, and a line starts at its first byte
            dst[0] = make_uint4(0u, 0u, 0u, 0u);
            F::classify_store(make_uint4(0u, 0u, 0u, 0u), bm16, 0u, bm_stride, term4);
            FL.dm16[0] = (uint16_t)fuse::pre_chunk_mask(0u, true);
        }
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            if ((uint32_t)k < nrow) stage(make_uint4(v[k][0], v[k][1], v[k][2], v[k][3]), k * kWave + lane + sh);
        }
        if (nrow > (uint32_t)NB) {  // rows beyond the register window (a format with a small window: LTSV, GELF with long lines)
            const uint32_t pre = T ? 0u : fuse::kPre;
            const uint64_t p0 = T ? g.base : 0ull;
            const uint64_t readable = padded - p0;
            const uint32_t want = g.span - pre;
            const uint32_t range = readable < (uint64_t)want ? (uint32_t)readable : want;
            __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(bytes + p0), (short)0, (int)range, 0x00020000);
            constexpr int TB = tail_batch<F>::value;
            for (uint32_t r0 = NB; r0 < nrow; r0 += TB) {
                u32x4 w[TB];
#pragma unroll
                for (int k = 0; k < TB; ++k) w[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(lane * 16u + (r0 + k) * 1024u), 0, FG_STREAM_AUX);
#pragma unroll
                for (int k = 0; k < TB; ++k)
                    if (r0 + k < nrow) stage(make_uint4(w[k][0], w[k][1], w[k][2], w[k][3]), (r0 + k) * kWave + lane + sh);
            }
        }
        if (nrow * kWave + sh < prev_nchunk)  // (stale masks of a longer tile before: only ever near the end of the stream)
            for (uint32_t c = nrow * kWave + sh + lane; c < prev_nchunk && c < dm_cap; c += kWave) FL.dm16[c] = 0u;
        prev_nchunk = nrow * kWave + sh;
        __builtin_amdgcn_sched_barrier(0);
        FG_MARK(A);
        __syncthreads();
        FG_CLK(9);
        if (g.end_x != fuse::kUnresolved) {  // the tile that holds the stream's end (one or two per launch): its masks again, out of LDS,
            err_acc = 0u;                    // with what lies behind the end read as zeros and the error bit AT the end kept
            const uint32_t* tw = reinterpret_cast<const uint32_t*>(smem);
            for (uint32_t c = lane; c < nchunk; c += kWave) {
                uint32_t m = fuse::chunk_masks(tw[c * 4u], tw[c * 4u + 1u], tw[c * 4u + 2u], tw[c * 4u + 3u], c ? tw[c * 4u - 1u] : 0u, term4, fuse::chunk_rem(g, c));
                if (c == 0u) m = fuse::pre_chunk_mask(m, T == 0u);
                err_acc |= m;
                FL.dm16[c] = (uint16_t)m;
            }
            __syncthreads();
        }
        // ---- the tile's lines: count, the tail that runs past the staged range ----
        fuse::Count cn = fuse::count_tile(FL, g);
        uint64_t tail_end = 0;
        bool tail_bad = false;
        if (cn.n_own != 0u && cn.n_all == cn.n_own) {  // wave-uniform: no terminator behind the last line's start in the staged range
            // Stage ON, a KiB at a time, while the tile has room: the line is then decoded out of LDS like every other (a line that is
            // not in the tile is parsed from global memory by ONE lane, ~0.3 ms while 63 wait: with a look-ahead of one average line
            // and no room to grow that was one tile in ten, and the whole kernel -- profiles/r06a_fused_frame.log).
            bool grown = false;
            FG_ST(4, 1);
            uint32_t carry = reinterpret_cast<const uint32_t*>(smem)[(g.span >> 2) - 1u];
            while (g.end_x == fuse::kUnresolved && g.span + 16u <= tile_cap) {
                const uint32_t ext = tile_cap - g.span < fa.ext ? tile_cap - g.span : fa.ext;  // bytes of this row (a multiple of 16, <= 1024)
                const uint64_t pos = g.base + g.span;                                        // (16-byte aligned, <= nbytes)
                if (pos + ext > fa.nbytes) g.end_x = (uint32_t)(fa.nbytes - g.base);         // the stream ends inside this row
                const uint32_t idx = (g.span >> 4) + lane;
                const uint64_t cpos = pos + (uint64_t)lane * 16u;
                const bool in_row = lane * 16u < ext;
                uint4 q = make_uint4(0u, 0u, 0u, 0u);
                if (in_row && cpos + 16u <= padded) q = stream_load(reinterpret_cast<const uint4*>(bytes + cpos));
                const uint32_t pw = wv::shfl_up1(q.w, carry);
                carry = (uint32_t)__builtin_amdgcn_readlane((int)q.w, 63);
                uint32_t m = 0u;
                if (in_row) {
                    dst[idx] = q;
                    F::classify_store(q, bm16, idx, bm_stride, term4);
                    m = fuse::chunk_masks(q.x, q.y, q.z, q.w, pw, term4, fuse::chunk_rem(g, idx));
                    err_acc |= m;
                    FL.dm16[idx] = (uint16_t)m;
                }
                g.span += ext;
                grown = true;
                if (__ballot((m & 0xFFFFu) != 0u) != 0ull) break;
            }
            if (grown) {
                if (g.end_x != fuse::kUnresolved) {  // the staged range now ends with the chunk that holds position nbytes
                    const uint32_t keep = (g.end_x + 16u) & ~15u;
                    for (uint32_t c = (keep >> 4) + lane; c < (g.span >> 4); c += kWave) FL.dm16[c] = 0u;
                    g.span = keep;
                }
                if ((g.span >> 4) > prev_nchunk) prev_nchunk = ((g.span >> 4) + kWave - 1u) & ~(kWave - 1u);
                __syncthreads();
                cn = fuse::count_tile(FL, g);
            }
            if (cn.n_all == cn.n_own) {  // still none: the line is longer than the tile, or the stream ends first
                uint64_t e = ~0ull;
                if (g.end_x == fuse::kUnresolved) {
                    FG_ST(5, 1);
                    const TailEnd te = fused_tail_scan(bytes, padded, fa.nbytes, term4, g.base + g.span, reinterpret_cast<const uint32_t*>(smem)[(g.span >> 2) - 1u]);
                    e = te.end;
                    tail_bad = te.bad != 0u;
                }
                if (e == ~0ull) {  // the stream ends first: a frame when the chunk is final (BufRead), else the caller's to carry over
                    if (fa.final_ != 0u) {
                        e = fa.nbytes;
                    } else {
                        cn.n_own -= 1u;
                        tail_bad = false;
                    }
                }
                tail_end = e;
            }
        }
        FG_CLK(10);
        const bool any_err = __ballot((err_acc >> 16) != 0u) != 0ull;  // wave-uniform; a healthy stream: never
        if (lane == 0u) {
            __hip_atomic_store(fa.tcount + T, cn.n_own + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            (void)__hip_atomic_fetch_add(fa.bagg + (T >> 6), (1ull << 40) | (unsigned long long)cn.n_own, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (any_err) FG_ST(7, 1);
        // ---- the tile before: its row index, then its rows.  BEFORE the prefetch is issued: vmcnt retires in order, so anything this
        // wave loads behind the window -- a poll of the look-back's slow path, a spilled register coming back -- waits for the whole
        // window (the first cut looked back behind the prefetch: 14 k of a tile's 68 k cycles sat right there, profiles/r06f_fused_stats.log).
        // (Measured and not kept: the next tile's ticket drawn at the top of the iteration instead of an iteration ahead -- fewer tiles wait on a
        // straggler, but the atomic sits in front of the look-back's loads: 0.98 -> 1.32 ms per GB, profiles/r06u_fused_frame.log.)
        if (pend_any) {
            const uint64_t base = lookback_finish(fa, pend_T, pend_nown, lbk, &aborted);
            if (aborted) return;
            store_rows(base, pend, pend_valid, pend_k, pend_o0, pend_o1);
            pend_any = false;
        }
        FG_CLK(12);
        // ---- prefetch: the next tile's bytes into the register window ----
        const uint64_t Tn = drawn();
        const bool more = Tn < fa.ntiles;  // wave-uniform
        if (more) {
            draw();
            load_window(Tn, fuse::tile_geo(Tn, fa.S, fa.look, fa.nbytes), v);
        }
        FG_CLK(11);
        __builtin_amdgcn_sched_barrier(0);
        FG_MARK(B);
        // ---- stage B: the tile's lines, L to the pass ----
        bool have_base = false;
        uint64_t base = 0;
        // (the rows of the tile before are stored: what the last pass leaves below is all that stays alive across decode())
        RowOut row{};
        uint64_t row_o0 = 0, row_o1 = 0;
        uint32_t row_k = 0;
        bool row_valid = false;
        const uint32_t* tile32 = reinterpret_cast<const uint32_t*>(smem);
        for (uint32_t w0 = 0; w0 < cn.n_own; w0 += fuse::kList) {
            __syncthreads();
            fuse::build_list(FL, cn, w0);
            __syncthreads();
            const uint32_t w1 = w0 + fuse::kList < cn.n_own ? w0 + fuse::kList : cn.n_own;
            for (uint32_t p0 = w0; p0 < w1; p0 += L) {
                if (p0 != 0u) {  // (a tile that holds more lines than a pass takes -- rare by the choice of S: the pass before goes out)
                    if (!have_base) {
                        base = fused_lookback(fa, T, cn.n_own, &aborted);
                        if (aborted) return;
                        have_base = true;
                    }
                    store_rows(base, row, row_valid, row_k, row_o0, row_o1);
                    __syncthreads();  // (this pass reads the tile again)
                }
                const uint32_t k = p0 + lane;
                const bool valid = lane < L && k < w1;
                // the lane's frame [o0, o1) (terminator included) from the list: asked for twice, before and BEHIND decode(), so that
                // nothing of it stays alive across the decoder (which runs at the register limit: a value kept there is a spill, and a
                // spilled register coming back is a wait for every load in flight)
                auto frame_of = [&](uint64_t* f0, uint64_t* f1, uint32_t* sx, uint32_t* ex) {
                    uint32_t s2, e2;
                    fuse::line_at(FL, cn, w0, k, &s2, &e2);
                    *f0 = g.base + s2;
                    *f1 = e2 == fuse::kUnresolved ? tail_end : g.base + e2;
                    *sx = s2, *ex = e2;
                };
                uint64_t o0 = 0, o1 = 0;
                bool bad = false;
                if (valid) {
                    uint32_t s, e;
                    frame_of(&o0, &o1, &s, &e);
                    if (e == fuse::kUnresolved) bad = tail_bad || (any_err && fused_line_bad(tile32, g.base, g.span, g.own_end, g.end_x, s, g.end_x != fuse::kUnresolved ? g.end_x + 1u : g.span));
                    else bad = any_err && fused_line_bad(tile32, g.base, g.span, g.own_end, g.end_x, s, e);
                }
                // terminator stripping (BufRead::lines / split(0), as persistent_loop)
                uint64_t e1 = o1;
                if (valid && e1 > o0) {
                    auto byte_at = [&](uint64_t q) -> uint32_t { return (q - g.base) < (uint64_t)g.span ? (uint32_t)smem[q - g.base] : (uint32_t)bytes[q]; };
                    const uint32_t b1 = byte_at(e1 - 1);
                    if (strip == FG_FRAME_LINE) {
                        if (b1 == '\n') {
                            --e1;
                            if (e1 > o0 && byte_at(e1 - 1) == '\r') --e1;
                        }
                    } else if (b1 == 0u) {
                        --e1;
                    }
                }
                GroupCtx c{bytes, smem, bm16, o0, e1, g.base, g.span, valid, 0ull, stash, 0u, ent_state, nullptr};
                c.tile_cap = tile_cap;
                FG_ST(6, 1);
                FG_MARK(C);
                row = fmt.decode(c, t);
                FG_MARK(D);
                if (valid && bad) {  // "Invalid UTF-8 input": the frame never reaches decode()
                    row.meta = FG_ST_BAD_UTF8 | (0xFFu << 8) | (0xFFu << 16);
                    row.ts = 0.0;
#pragma unroll
                    for (int q = 0; q < 6; ++q) row.span[q] = fg_span{0, FG_NONE};
                    row.count = 0;
                }
                row_k = p0 + lane;
                row_valid = lane < L && row_k < w1;
                row_o0 = row_o1 = 0;
                if (row_valid) {
                    uint32_t s3, e3;
                    frame_of(&row_o0, &row_o1, &s3, &e3);
                }
            }
        }
        // the rows of the tile's last pass wait for the tile's look-back, one iteration: by then every tile before has published
        // (a tile without a line of its own still looks back: it may be the one that publishes its block's prefix or the stream's total)
        if (have_base) {  // (the look-back is done: they can go now)
            store_rows(base, row, row_valid, row_k, row_o0, row_o1);
            row_valid = false;
        }
        pend = row;
        pend_o0 = row_o0, pend_o1 = row_o1, pend_k = row_k, pend_valid = row_valid;
        pend_any = !have_base;
        pend_T = T;
        pend_nown = cn.n_own;
        if (!more) break;
        __syncthreads();  // stage B's LDS reads are done before the next tile overwrites them
        T = Tn;
        g = fuse::tile_geo(T, fa.S, fa.look, fa.nbytes);
    }
    if (pend_any) {
        const uint64_t base = fused_lookback(fa, pend_T, pend_nown, &aborted);
        if (aborted) return;
        store_rows(base, pend, pend_valid, pend_k, pend_o0, pend_o1);
    }
#if defined(FG_FUSED_STATS)
    FG_CLK(13);
    if (lane == 0u)
        for (int k = 2; k < 14; ++k)
            if (k != 3) atomicAdd(fa.total + k, (unsigned long long)st_acc[k]);
#endif
}
#endif

// ---- host side: the grid and the kernel's arguments of a fused launch ----------------------------------------------------------------
// base_lds = the dynamic LDS the format's kernel needs for the tile g.tile without the fused block (as plan_launch adds it up).
// Zeroes the launch's scratch on `stream`.  Returns 0, or -1 (no device / a HIP error).
template <class K>
inline int fused_prepare(K kernel, const FusedGeom& g, uint32_t base_lds, uint64_t nbytes, int final_, uint32_t delim, uint64_t* d_offsets,
                         uint64_t cap, uint8_t* scratch, uint32_t stash_blocks, const fg_launch_opts& lo, hipStream_t stream, FusedArgs* fa,
                         uint32_t* lds_total, uint32_t* blocks) {
    fa->nbytes = nbytes;
    fa->S = g.S;
    fa->look = g.look;
    fa->delim4 = delim * 0x01010101u;
    fa->final_ = final_ ? 1u : 0u;
    fa->ext = g.ext >= 16u && g.ext <= 1024u ? g.ext & ~15u : 1024u;
    fa->cap = cap;
    fa->offsets = d_offsets;
    fused_carve(scratch, nbytes, g.S, fa);
    fa->lds_off = (base_lds + 15u) & ~15u;
    *lds_total = fa->lds_off + fuse::lds_bytes(g.tile);
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) return -1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kWave, *lds_total) != hipSuccess || per_cu < 1) per_cu = 1;
    if (lo.waves_per_cu > 0 && per_cu > (int)lo.waves_per_cu) per_cu = (int)lo.waves_per_cu;
    uint64_t b = (uint64_t)per_cu * (uint64_t)cus;
    if (stash_blocks && b > stash_blocks) b = stash_blocks;
    if (b > fa->ntiles) b = fa->ntiles;
    if (b < 1u) b = 1u;
    *blocks = (uint32_t)b;
#if defined(FG_FUSED_K)  // (A/B builds: fewer counters than the scratch has room for)
    constexpr uint32_t kUse = FG_FUSED_K < kFusedCounters ? FG_FUSED_K : kFusedCounters;
#else
    constexpr uint32_t kUse = kFusedCounters;
#endif
    fa->counters = (lo.flags & FG_LO_STATIC_CHUNKS) ? 0u : b < kUse ? (uint32_t)b : kUse;
    if (hipMemsetAsync(scratch, 0, fused_scratch_bytes(nbytes, g.S), stream) != hipSuccess) return -1;
    return 0;
}

}  // namespace fg
