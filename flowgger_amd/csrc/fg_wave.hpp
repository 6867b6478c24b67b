// fg_wave.hpp -- the wave-level vocabulary of the wave-cooperative tokenisers (fg_gelf2.hpp, fg_ltsv2.hpp, fg_sd2.hpp).
//
// Those tokenisers are written ONCE against the few primitives below and compiled twice:
//   * by hipcc for gfx950, where a primitive is one or two instructions (v_mbcnt, ds_bpermute, s_barrier, ds_*_rtn ...);
//   * by g++ for the CPU test suite, where the 64 lanes of a wave are 64 fibers and every cross-lane primitive is a
//     rendezvous (tests/native/fg_wave_emu.hpp) -- so `pytest -m "not gpu"` runs the very code the kernels run against the
//     oracle, lane for lane, before a GPU is involved.
// Execution model assumed by all users: ONE 64-lane wave per workgroup, cross-lane primitives are called by all 64 lanes in
// wave-uniform control flow (never under a divergent branch), LDS is private to the wave.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define FG_WV __device__ __forceinline__
#define FG_WVH __host__ __device__ inline  // geometry helpers the host launchers use as well
#else
#include <string.h>
#define FG_WV inline
#define FG_WVH inline
#include "fg_wave_emu.hpp"  // tests/native: the fiber scheduler behind the host versions below
#endif

// FG_MARK(k): with -DFG_ASM_MARKS a comment line "; FGMARK k" in the device assembly (tools/valu_count.py counts the instructions
// between marks: these kernels are VALU-issue bound, so the static count of a straight-line phase IS its cost); else nothing
#if defined(__HIP_DEVICE_COMPILE__) && defined(FG_ASM_MARKS)
#define FG_MARK(k) asm volatile("; FGMARK " #k)
#else
#define FG_MARK(k) ((void)0)
#endif

namespace fg {
namespace wv {

constexpr uint32_t kLanes = 64;

// ---------------------------------------------------------------------------------------------
// execution + cross-lane
// ---------------------------------------------------------------------------------------------
#if defined(__HIPCC__)
FG_WV uint32_t lane() { return threadIdx.x; }
FG_WV void sync() { __syncthreads(); }  // single-wave workgroup: orders LDS traffic between the lanes
FG_WV uint64_t ballot(bool p) { return __ballot(p); }
FG_WV uint32_t shfl(uint32_t v, uint32_t src) { return (uint32_t)__shfl((int)v, (int)src, 64); }
FG_WV uint32_t shfl_up(uint32_t v, uint32_t d) { return (uint32_t)__shfl_up((int)v, d, 64); }
// v of lane `src`, src WAVE-UNIFORM: one v_readlane instead of a trip through the LDS crossbar (ds_bpermute)
FG_WV uint32_t bcast(uint32_t v, uint32_t src) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, __builtin_amdgcn_readfirstlane((int)src));
}
// v of the lane below (lane 0 keeps `first`): DPP wave_shr:1
FG_WV uint32_t shfl_up1(uint32_t v, uint32_t first) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)first, (int)v, 0x138, 0xf, 0xf, false);
}
// number of set bits of m at positions below this lane
FG_WV uint32_t mbcnt(uint64_t m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
FG_WV uint32_t lds_or(uint32_t* p, uint32_t v) { return atomicOr(p, v); }
FG_WV uint32_t lds_xor(uint32_t* p, uint32_t v) { return atomicXor(p, v); }
FG_WV uint32_t lds_min(uint32_t* p, uint32_t v) { return atomicMin(p, v); }
FG_WV uint32_t lds_add(uint32_t* p, uint32_t v) { return atomicAdd(p, v); }
FG_WV unsigned long long glb_add(unsigned long long* p, unsigned long long v) { return atomicAdd(p, v); }
FG_WV uint32_t ctz32(uint32_t x) { return (uint32_t)__builtin_ctz(x); }
FG_WV uint32_t ctz64(uint64_t x) { return (uint32_t)__builtin_ctzll(x); }
FG_WV uint32_t clz64(uint64_t x) { return (uint32_t)__builtin_clzll(x); }
FG_WV uint32_t popc32(uint32_t x) { return (uint32_t)__builtin_popcount(x); }
FG_WV uint32_t popc64(uint64_t x) { return (uint32_t)__builtin_popcountll(x); }
FG_WV uint32_t alignbyte(uint32_t hi, uint32_t lo, uint32_t s) { return __builtin_amdgcn_alignbyte(hi, lo, s); }
FG_WV uint32_t udot4(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_udot4(a, b, c, false); }
FG_WV uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t s) { return __builtin_amdgcn_alignbit(hi, lo, s); }  // s in 0..31
FG_WV uint32_t mad24(uint32_t a, uint32_t b, uint32_t c) { return __umul24(a, b) + c; }  // a, b < 2^24
FG_WV uint32_t bfe(uint32_t v, uint32_t off, uint32_t width) { return __builtin_amdgcn_ubfe(v, off, width); }
// measurement builds: shader clock (after draining the LDS / memory queues so that a phase pays for what it started)
FG_WV uint64_t clock() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0xC07F);
    const uint64_t c = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
    return c;
}
#else
FG_WV uint32_t lane() { return emu::lane(); }
FG_WV void sync() { emu::collective(0, emu::OP_SYNC, 0); }
FG_WV uint64_t ballot(bool p) { return emu::collective(p ? 1u : 0u, emu::OP_BALLOT, 0); }
FG_WV uint32_t shfl(uint32_t v, uint32_t src) { return (uint32_t)emu::collective(v, emu::OP_SHFL, src & 63u); }
FG_WV uint32_t shfl_up(uint32_t v, uint32_t d) { return (uint32_t)emu::collective(v, emu::OP_SHFL_UP, d); }
FG_WV uint32_t bcast(uint32_t v, uint32_t src) { return (uint32_t)emu::collective(v, emu::OP_SHFL, src & 63u); }
FG_WV uint32_t shfl_up1(uint32_t v, uint32_t first) {
    const uint32_t r = (uint32_t)emu::collective(v, emu::OP_SHFL_UP, 1u);
    return emu::lane() == 0 ? first : r;
}
FG_WV uint32_t mbcnt(uint64_t m) { return (uint32_t)__builtin_popcountll(m & ((1ull << emu::lane()) - 1ull)); }
FG_WV uint32_t lds_or(uint32_t* p, uint32_t v) { uint32_t o = *p; *p = o | v; return o; }
FG_WV uint32_t lds_xor(uint32_t* p, uint32_t v) { uint32_t o = *p; *p = o ^ v; return o; }
FG_WV uint32_t lds_min(uint32_t* p, uint32_t v) { uint32_t o = *p; if (v < o) *p = v; return o; }
FG_WV uint32_t lds_add(uint32_t* p, uint32_t v) { uint32_t o = *p; *p = o + v; return o; }
FG_WV unsigned long long glb_add(unsigned long long* p, unsigned long long v) { unsigned long long o = *p; *p = o + v; return o; }
FG_WV uint32_t ctz32(uint32_t x) { return (uint32_t)__builtin_ctz(x); }
FG_WV uint32_t ctz64(uint64_t x) { return (uint32_t)__builtin_ctzll(x); }
FG_WV uint32_t clz64(uint64_t x) { return (uint32_t)__builtin_clzll(x); }
FG_WV uint32_t popc32(uint32_t x) { return (uint32_t)__builtin_popcount(x); }
FG_WV uint32_t popc64(uint64_t x) { return (uint32_t)__builtin_popcountll(x); }
FG_WV uint32_t alignbyte(uint32_t hi, uint32_t lo, uint32_t s) {
    const uint64_t v = ((uint64_t)hi << 32) | lo;
    return (uint32_t)(v >> (8u * (s & 3u)));
}
FG_WV uint32_t udot4(uint32_t a, uint32_t b, uint32_t c) {
    for (int k = 0; k < 4; ++k) c += ((a >> (8 * k)) & 0xFFu) * ((b >> (8 * k)) & 0xFFu);
    return c;
}
FG_WV uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t s) { return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> (s & 31u)); }
FG_WV uint32_t mad24(uint32_t a, uint32_t b, uint32_t c) { return (a & 0xFFFFFFu) * (b & 0xFFFFFFu) + c; }
FG_WV uint32_t bfe(uint32_t v, uint32_t off, uint32_t width) { return (v >> off) & ((1u << width) - 1u); }
FG_WV uint64_t clock() { return 0; }
#endif

FG_WV bool any(bool p) { return ballot(p) != 0ull; }

// wave-wide exclusive prefix sum; *total = the wave sum
#if defined(__HIPCC__)
// DPP scan (no LDS crossbar): three row shifts of the input, then shifts by 4 and 8 inside the rows of 16, then the two row
// broadcasts -- the classic GCN sequence.
template <int CTRL, int ROW_MASK, int BANK_MASK, bool BOUND>
FG_WV uint32_t dpp_(uint32_t src) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)src, CTRL, ROW_MASK, BANK_MASK, BOUND);
}
FG_WV uint32_t excl_sum(uint32_t v, uint32_t* total) {
    uint32_t x = v;
    x += dpp_<0x111, 0xf, 0xf, true>(v);   // row_shr:1
    x += dpp_<0x112, 0xf, 0xf, true>(v);   // row_shr:2
    x += dpp_<0x113, 0xf, 0xf, true>(v);   // row_shr:3
    x += dpp_<0x114, 0xf, 0xe, false>(x);  // row_shr:4, banks 1..3
    x += dpp_<0x118, 0xf, 0xc, false>(x);  // row_shr:8, banks 2..3
    x += dpp_<0x142, 0xa, 0xf, false>(x);  // row_bcast:15 into rows 1 and 3
    x += dpp_<0x143, 0xc, 0xf, false>(x);  // row_bcast:31 into rows 2 and 3
    *total = (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
    return x - v;
}
#else
FG_WV uint32_t excl_sum(uint32_t v, uint32_t* total) {
    const uint32_t l = lane();
    uint32_t inc = v;
    for (uint32_t d = 1; d < kLanes; d <<= 1) {
        const uint32_t t = shfl_up(inc, d);
        if (l >= d) inc += t;
    }
    *total = shfl(inc, kLanes - 1u);
    return inc - v;
}
#endif

// ---- rows of 16 lanes (the DPP row): rotations and all-to-all reductions without the LDS crossbar.  Every lane of the wave must be
//      active (a DPP read of a disabled lane returns nothing useful): wave-uniform control flow only.
#if defined(__HIPCC__)
template <int N>
FG_WV uint32_t row_ror(uint32_t v) {  // lane i of a row reads lane (i - N) mod 16 of the same row
    return dpp_<0x120 + N, 0xf, 0xf, true>(v);
}
#else
template <int N>
FG_WV uint32_t row_ror(uint32_t v) {
    const uint32_t l = lane();
    return shfl(v, (l & ~15u) | ((l - (uint32_t)N) & 15u));
}
#endif
// how many of the other fifteen lanes of this lane's row hold a value below x: the all-pairs count behind a rank inside a row.  The
// compiler keeps every rotation a separate v_mov_dpp (3 instructions a step; gfx950 has no DPP compare); spelled out, a subtract
// that reads its operand through DPP leaves "the other one is smaller" as its borrow and an add with carry-in collects it (2 a step).
#if defined(__HIP_DEVICE_COMPILE__)
FG_WV uint32_t row16_count_less(uint32_t x) {
    uint32_t acc = 0, tmp;
    // (s_nop 4: a VALU write of x, or of EXEC, just before a DPP read needs up to five wait states nobody adds inside an asm block)
#define FG_STEP_(N) "v_sub_co_u32_dpp %1, vcc, %2, %2 row_ror:" #N " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n\t"
    asm volatile("s_nop 4\n\t" FG_STEP_(1) FG_STEP_(2) FG_STEP_(3) FG_STEP_(4) FG_STEP_(5) FG_STEP_(6) FG_STEP_(7) FG_STEP_(8) FG_STEP_(9)
                     FG_STEP_(10) FG_STEP_(11) FG_STEP_(12) FG_STEP_(13) FG_STEP_(14) FG_STEP_(15)
                 : "+v"(acc), "=&v"(tmp)
                 : "v"(x)
                 : "vcc");
#undef FG_STEP_
    return acc;
}
#else
FG_WV uint32_t row16_count_less(uint32_t x) {
    uint32_t acc = 0;
    acc += row_ror<1>(x) < x ? 1u : 0u;
    acc += row_ror<2>(x) < x ? 1u : 0u;
    acc += row_ror<3>(x) < x ? 1u : 0u;
    acc += row_ror<4>(x) < x ? 1u : 0u;
    acc += row_ror<5>(x) < x ? 1u : 0u;
    acc += row_ror<6>(x) < x ? 1u : 0u;
    acc += row_ror<7>(x) < x ? 1u : 0u;
    acc += row_ror<8>(x) < x ? 1u : 0u;
    acc += row_ror<9>(x) < x ? 1u : 0u;
    acc += row_ror<10>(x) < x ? 1u : 0u;
    acc += row_ror<11>(x) < x ? 1u : 0u;
    acc += row_ror<12>(x) < x ? 1u : 0u;
    acc += row_ror<13>(x) < x ? 1u : 0u;
    acc += row_ror<14>(x) < x ? 1u : 0u;
    acc += row_ror<15>(x) < x ? 1u : 0u;
    return acc;
}
#endif
// OR / sum over the lanes of a row of (1 << lg) lanes, lg = 4, 5 or 6 (wave-uniform); every lane gets the result
FG_WV uint32_t rows_or(uint32_t x, uint32_t lg) {
    x |= row_ror<1>(x);
    x |= row_ror<2>(x);
    x |= row_ror<4>(x);
    x |= row_ror<8>(x);
    if (lg >= 5u) x |= shfl(x, lane() ^ 16u);
    if (lg >= 6u) x |= shfl(x, lane() ^ 32u);
    return x;
}
FG_WV uint32_t rows_add(uint32_t x, uint32_t lg) {
    x += row_ror<1>(x);
    x += row_ror<2>(x);
    x += row_ror<4>(x);
    x += row_ror<8>(x);
    if (lg >= 5u) x += shfl(x, lane() ^ 16u);
    if (lg >= 6u) x += shfl(x, lane() ^ 32u);
    return x;
}

// ---------------------------------------------------------------------------------------------
// Entry-table slots, wave-cooperative.  Every allocation used to be one atomic on the SAME global word (ent_used): 250 K atomics
// per 4 M lines serialise in one L2 channel and hold up everything else routed through it (measured: the loads of unrelated
// waves took tens of microseconds).  A wave now reserves a CHUNK and hands slots out of it; it goes back to the global word
// only when the chunk is used up.  Slots no line refers to (the end of a wave's last chunk) are legal: a line owns
// [ent_first, ent_first + ent_count), slices may lie anywhere in the table.
//   st     two words of the wave's LDS that persist across tiles: next free slot, slots left (zeroed at kernel start)
//   need   slots wanted (wave-uniform);  chunk  reservation size (wave-uniform)
// All 64 lanes call it together.
// ---------------------------------------------------------------------------------------------
// A request may be split at `cut` (a boundary between two lines' slices, chosen by the caller with wave_left()): slots [0, cut)
// come out of what is left of the current chunk, slots [cut, need) open a new one -- nothing but the tail of one line's slice is
// ever abandoned.  slot(i) = i < cut ? base0 + i : base1 + (i - cut).
struct Slots {
    uint32_t base0, cut, base1;
    bool overflow;
    FG_WV uint32_t at(uint32_t i) const { return i < cut ? base0 + i : base1 + (i - cut); }
};
FG_WV uint32_t wave_left(const uint32_t* st) { return st[1]; }
FG_WV Slots wave_alloc(unsigned long long* ent_used, uint64_t ent_cap, uint32_t* st, uint32_t need, uint32_t cut, uint32_t chunk) {
    Slots r{0u, need, 0u, false};
    if (need == 0u) return r;
    uint32_t next = st[0], left = st[1];
    sync();  // (everybody has read the state before lane 0 rewrites it)
    if (cut > need) cut = need;
    if (cut > left) cut = 0u;  // (a caller that did not look at wave_left(): everything goes to the new chunk)
    r.base0 = next;
    r.cut = cut;
    next += cut;
    left -= cut;
    const uint32_t rest = need - cut;
    if (rest != 0u) {
        const uint32_t grab = rest > chunk ? rest : chunk;
        unsigned long long b = 0;
        if (lane() == 0) b = glb_add(ent_used, (unsigned long long)grab);
        const uint32_t lo = bcast((uint32_t)b, 0u), hi = bcast((uint32_t)(b >> 32), 0u);
        b = ((unsigned long long)hi << 32) | lo;
        const unsigned long long room = b < ent_cap ? ent_cap - b : 0ull;
        const uint32_t usable = room < grab ? (uint32_t)room : grab;
        if (usable < rest) {
            // The table is full: the lines at and behind `cut` report FG_ST_OVERFLOW.  The lines before it KEEP their slots
            // [base0, base0 + cut), so that part of the chunk must be committed -- otherwise the next request that fits what is
            // left gets the same base0 and overwrites entries of rows that are not flagged (fg_hip.h: "tables are valid except
            // status == FG_ST_OVERFLOW rows").
            r.overflow = true;
            if (lane() == 0) {
                st[0] = next;
                st[1] = left;
            }
            sync();
            return r;
        }
        r.base1 = (uint32_t)b;
        next = (uint32_t)b + rest;
        left = usable - rest;
    }
    if (lane() == 0) {
        st[0] = next;
        st[1] = left;
    }
    sync();
    return r;
}
FG_WVH uint32_t alloc_chunk_for(uint64_t ent_cap, uint32_t waves, uint64_t n_lines = ~0ull) {
    // a wave strands what is left of its LAST chunk: keep the worst case (every wave, a whole chunk) below 1/16 of the table.
    // (Round 4: 1/64 and at most 1024 slots meant an atomic on ONE word every third group of the structured-data kernel -- 9 per
    // microsecond chip-wide, each waiting ~10 us in the queue of that word: 16 % of the group's time, profiles/r04g_phases_*.)
    // A table too small for chunks of 256 slots gets EXACT reservations (chunk 0: every request takes what it needs from the
    // global word and nothing is stranded) -- a caller that sized the table tightly must not see FG_ST_OVERFLOW because of
    // slots parked in chunks, and a table that small is not where the counter is contended.
    uint64_t c = ent_cap / (16ull * (waves ? waves : 1u));
    if (c > 4096u) c = 4096u;
    // ... and nothing like a whole chunk per wave when the LAUNCH is small: the sliced host paths decode one batch as dozens of
    // launches of ~60 K lines into one table, and every launch strands its own chunks (eight slots per line the wave will see)
    const uint64_t per_wave = n_lines / (waves ? waves : 1u) * 8ull;
    if (n_lines != ~0ull && c > per_wave) c = per_wave;
    if (c < 256u) c = 0u;
    return (uint32_t)c;
}

// ---------------------------------------------------------------------------------------------
// SWAR byte classes of a dword -> flags in bit 7 of each byte; four dwords of flags -> a 16-bit mask
// ---------------------------------------------------------------------------------------------
FG_WV uint32_t eq_flags(uint32_t x, uint32_t pat) {  // pat bytes < 0x80; exact
    const uint32_t s = ((x & 0x7F7F7F7Fu) ^ pat) + 0x7F7F7F7Fu;
    return ~(s | x) & 0x80808080u;
}
FG_WV uint32_t gt_flags(uint32_t x, uint32_t c) {  // byte > c (c < 0x7F); bytes >= 0x80 count as greater
    return (((x & 0x7F7F7F7Fu) + (0x7F7F7F7Fu - c * 0x01010101u)) | x) & 0x80808080u;
}
constexpr uint32_t kNoTerm = 0xFFFFFFFFu;    // "no frame terminator" for the classifiers
constexpr uint32_t kPastSpan = 0xFFFFFFFEu;  // "this chunk is the padding behind the staged bytes"
FG_WV uint32_t ctrl_flags(uint32_t x) {  // byte < 0x20
    const uint32_t ge32 = (x & 0x7F7F7F7Fu) + 0x60606060u;
    return ~(ge32 | x) & 0x80808080u;
}
FG_WV uint32_t gather16(uint32_t f0, uint32_t f1, uint32_t f2, uint32_t f3) {
    const uint32_t lo = udot4(f1, 0x80402010u, udot4(f0, 0x08040201u, 0u));
    const uint32_t hi = udot4(f3, 0x80402010u, udot4(f2, 0x08040201u, 0u));
    return (lo >> 7) | (hi << 1);
}

// ---------------------------------------------------------------------------------------------
// The wave's LDS tile: bytes as dwords, and bitmaps with one bit per tile byte (as dwords)
// ---------------------------------------------------------------------------------------------
struct Bytes {
    const uint32_t* w;
    FG_WV uint32_t byte(uint32_t a) const { return (w[a >> 2] >> (8u * (a & 3u))) & 0xFFu; }
    // 8 / 16 bytes starting at tile byte a (unaligned), little endian; the tile is padded: reading up to 19 bytes past
    // the last staged byte is safe
    FG_WV uint64_t load8(uint32_t a) const {
        const uint32_t d = a >> 2, s = a & 3u;
        const uint32_t w0 = w[d], w1 = w[d + 1], w2 = w[d + 2];
        return (uint64_t)alignbyte(w1, w0, s) | ((uint64_t)alignbyte(w2, w1, s) << 32);
    }
    FG_WV void load16(uint32_t a, uint32_t q[4]) const {
        const uint32_t d = a >> 2, s = a & 3u;
        const uint32_t r0 = w[d], r1 = w[d + 1], r2 = w[d + 2], r3 = w[d + 3], r4 = w[d + 4];
        q[0] = alignbyte(r1, r0, s);
        q[1] = alignbyte(r2, r1, s);
        q[2] = alignbyte(r3, r2, s);
        q[3] = alignbyte(r4, r3, s);
    }
    FG_WV void load24(uint32_t a, uint32_t q[6]) const {  // (reads up to 27 bytes past a: inside the tile's padding)
        const uint32_t d = a >> 2, s = a & 3u;
        uint32_t r[7];
#pragma unroll
        for (uint32_t k = 0; k < 7u; ++k) r[k] = w[d + k];
#pragma unroll
        for (uint32_t k = 0; k < 6u; ++k) q[k] = alignbyte(r[k + 1], r[k], s);
    }
};

// first set bit of bitmap bm at tile position >= p and < lim, else lim.  INV = true searches the complement.
// (bitmaps are readable one dword past the tile's last bit word)
template <bool INV = false>
FG_WV uint32_t find_bit(const uint32_t* bm, uint32_t p, uint32_t lim) {
    while (p < lim) {
        uint32_t w = bm[p >> 5];
        if (INV) w = ~w;
        w >>= (p & 31u);
        if (w) {
            const uint32_t r = p + ctz32(w);
            return r < lim ? r : lim;
        }
        p = (p | 31u) + 1u;
    }
    return lim;
}
// 64 bits of bitmap bm starting at tile position a (bit i <=> tile byte a + i); reads three dwords
FG_WV uint64_t window64(const uint32_t* bm, uint32_t a) {
    const uint32_t d = a >> 5, s = a & 31u;
    const uint32_t w0 = bm[d], w1 = bm[d + 1], w2 = bm[d + 2];
    return (uint64_t)alignbit(w1, w0, s) | ((uint64_t)alignbit(w2, w1, s) << 32);
}
// any bit of bm set in [a, b)?
FG_WV bool any_bit(const uint32_t* bm, uint32_t a, uint32_t b) { return find_bit<false>(bm, a, b) < b; }

}  // namespace wv
}  // namespace fg
