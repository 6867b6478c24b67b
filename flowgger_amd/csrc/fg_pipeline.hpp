// fg_pipeline.hpp -- the HBM -> VGPR -> LDS streaming skeleton shared by the three decode kernels,
// plus the byte-class / bitmap / window primitives their lane-per-line tokenisers are built from.
//
// Execution model (gfx950, 64-lane waves, one wave per workgroup, PERSISTENT grid):
//   * a wave walks line groups g = blockIdx.x, +gridDim.x, ... ; a group = L consecutive lines
//     (L = 64 for ~256-byte lines, smaller powers of two for longer lines) = one contiguous byte
//     range of the packed buffer;
//   * stage A: the group's bytes arrive in NB x 16 B of VGPRs per lane (a register prefetch
//     window filled by bounds-checked buffer loads while the PREVIOUS group was being tokenised),
//     are classified while they sit in registers (one 16-bit mask per 16-byte chunk -> a bitmap
//     of the format's delimiter) and written to the wave's LDS tile;
//   * stage B: every lane tokenises ITS line out of LDS (format-specific, F::decode);
//   * the table row of a group is stored one iteration late (see the comment in the loop).
// HBM sees only coalesced 1 KiB-per-wave-instruction reads, each byte once.
#pragma once
#include "fg_device.hpp"
#include "fg_plan_policy.hpp"
#include "fg_wave.hpp"

namespace fg {

using u32x4 = unsigned int __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------
// SWAR byte classes -> 16-bit masks
// ---------------------------------------------------------------------------------------------
// bit7 of each byte set <=> byte == pat's byte (pat bytes < 0x80); exact, no borrow artefacts
__device__ __forceinline__ uint32_t eq_flags(uint32_t x, uint32_t pat) {
    uint32_t s = ((x & 0x7F7F7F7Fu) ^ pat) + 0x7F7F7F7Fu;  // bit7 set <=> low 7 bits != pat
    return ~(s | x) & 0x80808080u;
}
// bit7 of each byte set <=> 33 <= byte <= 126
__device__ __forceinline__ uint32_t range_flags(uint32_t x) {
    uint32_t l = x & 0x7F7F7F7Fu;
    uint32_t ge33 = l + 0x5F5F5F5Fu;   // bit7 set <=> low7 >= 33
    uint32_t le126 = l + 0x01010101u;  // bit7 set <=> low7 == 127
    return ge33 & ~le126 & ~x & 0x80808080u;
}
// bit7 of each byte set <=> byte < 0x20 (control characters)
__device__ __forceinline__ uint32_t ctrl_flags(uint32_t x) {
    uint32_t ge32 = (x & 0x7F7F7F7Fu) + 0x60606060u;  // bit7 set <=> low7 >= 32
    return ~(ge32 | x) & 0x80808080u;
}
// four dwords of byte flags (0x80 per hit) -> 16-bit mask, v_dot4_u32_u8 as the bit gather
__device__ __forceinline__ uint32_t gather16(uint32_t f0, uint32_t f1, uint32_t f2, uint32_t f3) {
    uint32_t lo = __builtin_amdgcn_udot4(f1, 0x80402010u, __builtin_amdgcn_udot4(f0, 0x08040201u, 0u, false), false);
    uint32_t hi = __builtin_amdgcn_udot4(f3, 0x80402010u, __builtin_amdgcn_udot4(f2, 0x08040201u, 0u, false), false);
    return (lo >> 7) | (hi << 1);
}
__device__ __forceinline__ uint32_t mask16_eq(const uint4& v, uint32_t pat) {
    return gather16(eq_flags(v.x, pat), eq_flags(v.y, pat), eq_flags(v.z, pat), eq_flags(v.w, pat));
}

// ---------------------------------------------------------------------------------------------
// The wave's LDS tile: bytes (as dwords) + one bitmap (bit i <=> tile byte i is in the class)
// ---------------------------------------------------------------------------------------------
struct Tile {
    const uint32_t* w;
    const uint32_t* bm;
};
// 8 / 16 consecutive bytes starting at tile byte `a` (unaligned) as little-endian dwords
__device__ __forceinline__ void load8(const Tile& T, uint32_t a, uint32_t* lo, uint32_t* hi) {
    uint32_t d = a >> 2, s = a & 3u;
    uint32_t w0 = T.w[d], w1 = T.w[d + 1], w2 = T.w[d + 2];
    *lo = __builtin_amdgcn_alignbyte(w1, w0, s);
    *hi = __builtin_amdgcn_alignbyte(w2, w1, s);
}
__device__ __forceinline__ void load16(const Tile& T, uint32_t a, uint32_t w[4]) {
    const uint32_t d = a >> 2, s = a & 3u;
    uint32_t r0 = T.w[d], r1 = T.w[d + 1], r2 = T.w[d + 2], r3 = T.w[d + 3], r4 = T.w[d + 4];
    w[0] = __builtin_amdgcn_alignbyte(r1, r0, s);
    w[1] = __builtin_amdgcn_alignbyte(r2, r1, s);
    w[2] = __builtin_amdgcn_alignbyte(r3, r2, s);
    w[3] = __builtin_amdgcn_alignbyte(r4, r3, s);
}
// Byte reader over a line in the tile that keeps 16 bytes in registers: a short forward walk
// (a number, a timestamp) pays ONE LDS round trip instead of one per dword.  Same interface as
// LdsReader (byte(i), i = line-relative index).
struct WinReader {
    const Tile& T;
    uint32_t base;
    uint32_t w0 = 0x80000000u;  // line index of the window's first byte (initially: no index is within 16 of it)
    uint64_t lo = 0, hi = 0;
    __device__ __forceinline__ WinReader(const Tile& t, uint32_t b) : T(t), base(b) {}
    __device__ __forceinline__ uint32_t byte(uint32_t i) {
        uint32_t off = i - w0;
        if (off >= 16u) {
            uint32_t w[4];
            load16(T, base + i, w);
            lo = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
            hi = (uint64_t)w[2] | ((uint64_t)w[3] << 32);
            w0 = i;
            off = 0;
        }
        const uint64_t v = off < 8u ? lo : hi;
        return (uint32_t)(v >> (8u * (off & 7u))) & 0xFFu;
    }
};

// first set bit of a tile bitmap at line index >= q (tile byte base+q), or len; 32 bytes per step
// (compact: for short fields -- names, keys, SD values)
__device__ __forceinline__ uint32_t find_bit(const uint32_t* bm, uint32_t base, uint32_t q, uint32_t len) {
    while (q < len) {
        uint32_t a = base + q;
        uint32_t w = bm[a >> 5] >> (a & 31u);
        if (w) {
            uint32_t r = q + (uint32_t)__builtin_ctz(w);
            return r < len ? r : len;
        }
        q += 32u - (a & 31u);
    }
    return len;
}
// The same for spans that may be long (an 8 KiB message, a JSON text field): the first word is
// handled alone (the usual hit), a long run without the class is then crossed 256 bytes per
// step, eight independent LDS reads per round trip.
__device__ __forceinline__ uint32_t find_bit_long(const uint32_t* bm, uint32_t base, uint32_t q, uint32_t len) {
    if (q >= len) return len;
    {
        const uint32_t a = base + q;
        const uint32_t w = bm[a >> 5] >> (a & 31u);
        if (w) {
            const uint32_t r = q + (uint32_t)__builtin_ctz(w);
            return r < len ? r : len;
        }
        q += 32u - (a & 31u);
    }
    while (q + 256u <= len) {  // base + q is a multiple of 32 from here on
        const uint32_t i = (base + q) >> 5;
        uint32_t w[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) w[k] = bm[i + k];
        if ((w[0] | w[1] | w[2] | w[3] | w[4] | w[5] | w[6] | w[7]) != 0u) {
            uint32_t r = len;
#pragma unroll
            for (int k = 7; k >= 0; --k)
                if (w[k]) r = q + 32u * k + (uint32_t)__builtin_ctz(w[k]);
            return r < len ? r : len;
        }
        q += 256u;
    }
    while (q < len) {
        const uint32_t w = bm[(base + q) >> 5];
        if (w) {
            const uint32_t r = q + (uint32_t)__builtin_ctz(w);
            return r < len ? r : len;
        }
        q += 32u;
    }
    return len;
}
// wave-cooperative: rebuild the bitmap of the staged tile for another byte class (M::mask16)
template <class M>
__device__ __forceinline__ void rebuild_bitmap(const uint8_t* smem, uint16_t* bm16, uint32_t nchunk) {
    const uint4* src = reinterpret_cast<const uint4*>(smem);
    for (uint32_t c = threadIdx.x; c < nchunk; c += kWave) bm16[c] = (uint16_t)M::mask16(src[c]);
}

// ---------------------------------------------------------------------------------------------
// Output row (the fixed 68 bytes of a line) and the per-wave entry stash
// ---------------------------------------------------------------------------------------------
struct RowOut {
    uint32_t meta;
    double ts;
    fg_span span[6];
    uint32_t first, count;
    bool skip = false;  // the format has written what it wants of this row itself (kernels that complete rows of an earlier kernel)
};
__device__ __forceinline__ void store_row(const DevTables& t, uint64_t li, const RowOut& o) {
    if (o.skip) return;
    gstore(t.meta, li, o.meta);
    gstore(t.ts, li, o.ts);
#pragma unroll
    for (int k = 0; k < 6; ++k) gstore(t.span[k], li, o.span[k]);
    gstore(t.ent_first, li, o.first);
    gstore(t.ent_count, li, o.count);
}
#if defined(FG_ROW64_AB)
// MEASUREMENT VARIANT ONLY (FG_BUILD_VARIANT=row64a / row64b, tools/probe/row64_ab.py): the fixed part of a line as
// ONE 64-byte record (meta, ts, six spans, entry count -- the entry index folded away) at (uint4*)t.meta + 4 * line, instead of nine
// column stores.  The tables of this build are NOT the ABI's: nothing but the timing of the decode launch means anything.
//   FG_ROW64_AB == 1  each lane stores its own record: four 16-byte stores at a 64-byte stride
//   FG_ROW64_AB == 2  the wave's 64 records transposed through the (free) head of the LDS tile: four stores of 1 KiB contiguous each
__device__ __forceinline__ void pack_row64(const RowOut& o, uint4 w[4]) {
    const uint64_t tsb = (uint64_t)__double_as_longlong(o.ts);
    w[0] = make_uint4(o.meta, (uint32_t)tsb, (uint32_t)(tsb >> 32), o.span[0].off);
    w[1] = make_uint4(o.span[0].len, o.span[1].off, o.span[1].len, o.span[2].off);
    w[2] = make_uint4(o.span[2].len, o.span[3].off, o.span[3].len, o.span[4].off);
    w[3] = make_uint4(o.span[4].len, o.span[5].off, o.span[5].len, o.count);
}
__device__ __forceinline__ void store_row64_lane(const DevTables& t, uint64_t li, const RowOut& o) {
    if (o.skip) return;
    uint4 w[4];
    pack_row64(o, w);
    uint4* g = reinterpret_cast<uint4*>(t.meta) + li * 4u;
#pragma unroll
    for (int j = 0; j < 4; ++j) g[j] = w[j];
}
#endif
// Entries found while a line is parsed are parked in the wave's scratch (stash[k * 64 + lane],
// k < kStashEntries, two u64 per entry in the GELF/LTSV kernels, one in the RFC5424 kernel) and
// copied into the entry table once the wave has its slots -- instead of parsing every line twice.
constexpr uint32_t kStashEntries = 48;
constexpr uint32_t kStashWords = 2;  // u64 words per stashed entry (scratch is sized for this)

// Wave-aggregated allocation of entry slots out of the wave's reserved chunk (fg_wave.hpp wave_alloc): returns this lane's
// first slot, or sets *overflow.  ent_state = the wave's two persistent LDS words.
struct EntAlloc {
    uint32_t ex = 0;      // entries of the lanes before this one
    uint32_t total = 0;   // entries of the wave
    wv::Slots s{};        // where they go: [base0, base0 + cut) and [base1, ...) (wv::wave_alloc)
    bool overflow = false;  // THIS lane's entries did not get slots (the table is full)
};
__device__ __forceinline__ EntAlloc alloc_entries_ex(const DevTables& t, uint32_t n_ent, uint32_t* ent_state) {
    EntAlloc a;
    if (!__any(n_ent != 0u)) return a;  // wave-uniform; the common case of the no-SD corpus skips the scan
    a.ex = wv::excl_sum(n_ent, &a.total);
    // lines whose slice still fits what is left of the wave's chunk stay there, the others open the next chunk
    const uint32_t left = wv::wave_left(ent_state);
    const unsigned long long nofit = __ballot(n_ent != 0u && a.ex + n_ent > left);
    const uint32_t cut = nofit ? (uint32_t)__shfl((int)a.ex, (int)__builtin_ctzll(nofit), kWave) : a.total;
    a.s = wv::wave_alloc(t.ent_used, t.ent_cap, ent_state, a.total, cut, t.alloc_chunk);
    a.overflow = a.s.overflow && n_ent != 0u && a.ex >= a.s.cut;
    return a;
}
__device__ __forceinline__ uint32_t alloc_entries(const DevTables& t, uint32_t n_ent, bool* overflow, uint32_t* ent_state) {
    const EntAlloc a = alloc_entries_ex(t, n_ent, ent_state);
    *overflow = a.overflow;
    return (a.overflow || a.total == 0u) ? 0u : a.s.at(a.ex);
}

// ---------------------------------------------------------------------------------------------
// The persistent streaming loop.  F supplies
//     static constexpr uint32_t F::kClasses                                    stage-A class bitmaps kept in LDS (0..5)
//     static void F::classify_store(const uint4& q, uint16_t* bm16, uint32_t chunk, uint32_t stride, uint32_t term4)
//         (term4: the framing's terminator byte in all four bytes of a dword, wv::kNoTerm without framing,
//          wv::kPastSpan for a chunk of zeros behind the last staged byte)
//                                                                              16-bit class masks of a 16-byte chunk -> bm16[c * stride + chunk]
//     RowOut F::decode(const GroupCtx&)  (member, may use its own state; called by all 64 lanes, converged)
// ---------------------------------------------------------------------------------------------
struct GroupCtx {
    const uint8_t* bytes;   // packed buffer (global)
    const uint8_t* smem;    // tile bytes
    uint16_t* bm16;         // tile bitmap (stage A's class; a decoder may rebuild it)
    uint64_t o0, o1;        // this lane's line [o0, o1) in the packed buffer
    uint64_t a0;            // packed-buffer address of tile byte 0
    uint32_t span;          // tile bytes staged
    bool valid;             // this lane owns a line
    uint64_t li;            // its index in the batch
    uint64_t* stash;        // the wave's entry stash (or null)
    uint32_t ablate;        // measurement build only
    uint32_t* ent_state;    // the wave's entry-slot reservation (two LDS words that persist across groups, wv::wave_alloc)
    unsigned long long* phase;  // measurement build only: ten format-specific phase clocks (lane 0 adds), else null
    // HEAD staging (persistent_loop<..., HEAD = true>; else unused): the tile holds only the first bytes of every line
    uint32_t tbase = 0;     // tile byte of this lane's line
    uint32_t tlen = 0;      // bytes of the line that are in the tile, from its first byte (== its length when it is there whole)
    uint32_t last2 = 0;     // the line's last byte | the byte before it << 8 (the trims look there; 0 where the line has none)
    uint32_t tile_cap = 0;  // LDS tile bytes of this launch (the bitmaps and a format's extra block lie behind it)
};

// The entries a wave parked in its stash -> the entry table THROUGH THE TILE: once every lane has parsed its line the tile's
// bytes are dead, so the records are laid out there in slot order and leave with one entry per lane and store instruction --
// consecutive lanes, consecutive addresses (lane-per-line stores are `n_ent` entries apart: every store instruction touched
// every cache line of the wave's range, and the 1-byte columns were read-modify-written at HBM: 2.2x the algorithmic bytes).
// WORDS = u64 words per stashed record (k-major: stash[(k * WORDS + w) * 64 + lane]); un(w0, w1, &name, &val, &type_flags).
// mine = this lane's entries are in the stash.  false = not done (the caller stores lane by lane): some lane's entries are
// not in the stash, the table overflowed, or the wave has more entries than fit the tile.
template <uint32_t WORDS, class Unpack>
__device__ __forceinline__ bool stash_to_table(const GroupCtx& c, const DevTables& t, const EntAlloc& a, uint32_t n_ent, bool mine,
                                               Unpack un) {
    const uint32_t lane = threadIdx.x;
    const uint32_t cap = (c.span / 18u) & ~3u;  // 8 + 8 + 2 bytes per entry
    if (a.total == 0u || a.total > cap || a.s.overflow || c.stash == nullptr || __any(n_ent != 0u && !mine)) return false;  // wave-uniform
    uint64_t* s_name = reinterpret_cast<uint64_t*>(const_cast<uint8_t*>(c.smem));
    uint64_t* s_val = s_name + cap;
    uint16_t* s_tf = reinterpret_cast<uint16_t*>(s_val + cap);
    __syncthreads();  // every lane is done with the tile's bytes
    for (uint32_t k0 = 0; k0 < n_ent; k0 += 4u) {  // (four records in flight per lane)
        uint64_t w0[4], w1[4];
#pragma unroll
        for (uint32_t j = 0; j < 4u; ++j) {
            const uint32_t k = k0 + j < n_ent ? k0 + j : n_ent - 1u;
            w0[j] = c.stash[(k * WORDS) * kWave + lane];
            w1[j] = WORDS > 1u ? c.stash[(k * WORDS + 1u) * kWave + lane] : 0ull;
        }
#pragma unroll
        for (uint32_t j = 0; j < 4u; ++j) {
            if (k0 + j < n_ent) {
                uint64_t name, val;
                uint32_t tf;
                un(w0[j], w1[j], &name, &val, &tf);
                const uint32_t idx = a.ex + k0 + j;
                s_name[idx] = name;
                s_val[idx] = val;
                s_tf[idx] = (uint16_t)tf;
            }
        }
    }
    __syncthreads();
    for (uint32_t idx = lane; idx < a.total; idx += kWave) {
        const uint32_t slot = a.s.at(idx);
        const uint64_t name = s_name[idx];
        const uint32_t tf = s_tf[idx];
        t.ent_name[slot] = fg_span{(uint32_t)name, (uint32_t)(name >> 32)};
        t.ent_val[slot] = s_val[idx];
        t.ent_type[slot] = (uint8_t)(tf & 0xFFu);
        t.ent_flags[slot] = (uint8_t)(tf >> 8);
    }
    return true;
}


// Framing of the frames handed to the decoders (fg_decode_frames_device): what to strip from the
// end of [offsets[i], offsets[i+1]) before decoding, and which frames to reject as invalid UTF-8.
struct FrameArgs {
    uint32_t strip;           // FG_FRAME_NONE / _LINE ("\n", then one "\r") / _NUL ("\0")
    const uint8_t* line_bad;  // [n] 1 = not valid UTF-8 (or null)
    // Dynamic chunk dispatch (round 5): the launch's ticket counter (one word of a ctx-owned ring) and the value it holds when the
    // launch starts; null = chunks are dealt out round-robin (the form of rounds 3-4, kept for A/B: FG_LO_STATIC_CHUNKS).
    uint32_t* ticket = nullptr;
    uint32_t ticket_base = 0;
    // ... and the chunk indices from which the chunks of a ticket launch shrink (fg_plan_policy.hpp: the taper)
    uint32_t taper0 = kNoTaper, taper1 = kNoTaper, taper2 = kNoTaper;
};
// (host side of the ticket: fg::TicketSlot, fg_tables_view.hpp)

// The input is read ONCE: its loads carry the non-temporal hint (aux bit 1 = nt on gfx950) so that the stream does not push the
// wave's stash, scratch and table lines out of the XCD's L2 between two uses.
#ifndef FG_STREAM_AUX
#define FG_STREAM_AUX 2
#endif
__device__ __forceinline__ uint4 stream_load(const uint4* p) {
#if FG_STREAM_AUX
    typedef uint32_t nt4 __attribute__((ext_vector_type(4)));
    const nt4 v = __builtin_nontemporal_load(reinterpret_cast<const nt4*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
#else
    return *p;
#endif
}

// rows of 1 KiB the tail of stage A loads before it stores the first (a format overrides it with  static constexpr int kTailBatch)
template <class F, class = void>
struct tail_batch { static constexpr int value = 4; };
template <class F>
struct tail_batch<F, decltype((void)F::kTailBatch)> { static constexpr int value = F::kTailBatch; };

// The table row of a group is normally stored one iteration late (see below); a format says  static constexpr bool kDeferRowStore =
// false  when that only keeps 17 registers alive across stage A (and, at the register limit, in scratch memory)
template <class F, class = void>
struct defer_row_store { static constexpr bool value = true; };
template <class F>
struct defer_row_store<F, decltype((void)F::kDeferRowStore)> { static constexpr bool value = F::kDeferRowStore; };

// HEAD = true: LONG lines.  Per-line parse work does not grow with the message, and the lines a CU can have in flight are what its LDS
// holds (these kernels run at lines-in-flight / per-line latency: tools/sweep.py, throughput linear in the waves per CU) -- so a
// format that never looks inside the message (RFC5424: header, structured data, the trims) stages only the HEAD of every line,
// its first kHeadCap bytes, one 1 KiB row per line (coalesced, bounds-checked by a buffer descriptor per row): three to four times
// the lines per tile on the 64 B .. 8 KiB corpus.  The decoder gets the line's last two bytes beside it (the trims) and takes the
// rare line whose structured data runs past its head from global memory.
constexpr uint32_t kHeadCap = 1024;  // bytes of a line staged in HEAD mode, alignment slack included (a multiple of 16)
// (a format whose head holds less says so -- F::kHeadBytes, a multiple of 16 below kHeadCap: more lines per tile at the same occupancy)
template <class F, class = void>
struct head_cap { static constexpr uint32_t value = kHeadCap; };
template <class F>
struct head_cap<F, decltype((void)F::kHeadBytes)> { static constexpr uint32_t value = F::kHeadBytes; };

template <int NB, bool PROF, class F, bool HEAD = false>
__device__ __forceinline__ void persistent_loop(const uint8_t* __restrict__ bytes, const uint64_t* __restrict__ offsets,
                                                uint64_t n, const DevTables& t, uint32_t tile_cap, uint32_t L, uint64_t chunk_lines,
                                                unsigned long long* prof, uint64_t* stash_base, F& fmt, FrameArgs fr) {
    static_assert(!HEAD || F::kClasses <= 1, "HEAD staging classifies one stage-A byte class at most");
    const uint64_t kChunkLines = chunk_lines;  // lines a wave takes at a time (LaunchPlan::chunk)
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint16_t* bm16 = reinterpret_cast<uint16_t*>(smem + tile_cap + 64u);
    const uint32_t bm_stride = tile_cap / 16u + 16u;  // u16 entries per class bitmap (F::kClasses of them, back to back)
    uint4* dst = reinterpret_cast<uint4*>(smem);
    const uint32_t lane = threadIdx.x;
    const uint64_t G = gridDim.x;
    uint64_t* stash = stash_base ? stash_base + (uint64_t)blockIdx.x * (kStashEntries * kStashWords * kWave) : nullptr;

    // A wave owns a CONTIGUOUS range of lines [p, hi) and cuts it into groups BY BYTES: a group = as many consecutive lines as
    // fit the tile, at most L of them (one per lane).  (Round 2 cut groups by line count, L lines of AVERAGE length per tile: on
    // the long-tail corpora -- 64 B .. 8 KiB -- a third of the groups overflowed the tile and were finished in further passes over
    // a synchronously restaged tile, which is where those workloads spent their time.)  Short lines give groups of L lines as
    // before; only a single line longer than the whole tile is still parsed from global memory.
    // The ranges are CHUNKS of kChunkLines lines dealt out round-robin (wave b takes chunks b, b + G, ...): the grid as a whole
    // still sweeps the buffer front to back.  (One big range per wave -- 1792 streams spread over 25 GB -- cost the headline
    // configuration 10 %: DRAM pages and TLB entries want neighbours in time to be neighbours in memory.)  The last group of a
    // chunk may be short: 256 lines = 4 full groups of 64 short lines, a dozen or more groups of long ones (small batches get
    // smaller chunks, so that every wave of the grid has one).
    // Round 5: only a wave's FIRST chunk is its block index; every further one is DRAWN from the launch's ticket counter (one atomic
    // per chunk, by lane 0, requested when the chunk BEFORE it is entered, so that a whole chunk hides its round trip).
    // Tickets are handed out in order, so the sweep stays front to back -- and a wave that meets a slow line (a walk through global
    // memory), a slow XCD or a chunk of long lines simply draws fewer tickets instead of keeping the whole grid waiting at the end:
    // what a batch of 10^5 .. 10^6 lines, one or two chunks per wave, needs.  Every wave draws until its first
    // ticket beyond the last chunk, so a launch advances the counter by EXACTLY the number of chunks: the host keeps the count and
    // hands the next launch of the slot its base -- no reset, no memset between launches.
    const bool dyn = fr.ticket != nullptr;  // wave-uniform
    uint32_t tk_raw = 0;      // the ticket drawn ahead (lane 0's return value; read when the chunk in hand runs out)
    bool tk_pending = false;  // wave-uniform
    auto draw_ticket = [&]() {
        if (lane == 0u) tk_raw = __hip_atomic_fetch_add(fr.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tk_pending = true;
    };
    uint64_t chunk = blockIdx.x;
    uint64_t p, hi_line;
    chunk_range(chunk, kChunkLines, fr.taper0, fr.taper1, fr.taper2, n, &p, &hi_line);
    if (dyn && p < n) draw_ticket();

    // the L offsets from line q on (clamped to the wave's range), as o0 = start / o1 = end of lane's line
    auto load_offsets = [&](uint64_t q, uint64_t* o0, uint64_t* o1) {
        const uint64_t li = q + lane;
        const bool in = lane < L && li < hi_line;
        *o0 = offsets[in ? li : hi_line];
        *o1 = offsets[in ? li + 1 : hi_line];
    };
    // group geometry from the lanes' offsets, as SCALARS: lines in the group, tile start (16-byte aligned), staged span
    auto geometry = [&](uint64_t q, uint64_t o0, uint64_t o1, uint32_t* nl, uint64_t* a0, uint32_t* span) {
        const uint64_t left = hi_line - q;
        const uint32_t avail = left < (uint64_t)L ? (uint32_t)left : L;  // >= 1
        // (the builtins return int: widen through uint32_t or bit 31 sign-extends into the high word)
        const uint32_t lo_l = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)o0);
        const uint32_t lo_h = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(o0 >> 32));
        const uint64_t lo = (uint64_t)lo_l | ((uint64_t)lo_h << 32);
        *a0 = lo & ~15ull;
        // leading lanes whose line ends inside the tile; the first line always goes (alone, from global memory, when too long)
        const unsigned long long fit = __ballot(lane < avail && (o1 - *a0) <= (uint64_t)tile_cap);
        uint32_t cnt = fit == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~fit);  // leading ones
        if (cnt == 0u) cnt = 1u;
        *nl = cnt;
        const uint32_t last = cnt - 1u;
        const uint32_t hi_l = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)o1, (int)last);
        const uint32_t hi_h = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(o1 >> 32), (int)last);
        const uint64_t hi = (uint64_t)hi_l | ((uint64_t)hi_h << 32);
        const uint64_t want = hi - *a0;
        *span = want > tile_cap ? tile_cap : (uint32_t)((want + 15ull) & ~15ull);
    };
    // the register window: NB buffer loads of 16 B per lane; the buffer descriptor bounds the
    // tile, so rows past `span` fetch nothing and return zeros -- no per-row predication.  The
    // row offset goes into the VGPR/immediate offset (the part the hardware range-checks; the
    // scalar offset is not checked).
    auto load_window = [&](uint64_t a0, uint32_t span, u32x4* v) {
        __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(bytes + a0), (short)0, (int)span, 0x00020000);
#pragma unroll
        for (int k = 0; k < NB; ++k) v[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(lane * 16u + k * 1024u), 0, FG_STREAM_AUX);
    };
    // ---- HEAD mode ----
    // per lane: st = bytes of its line's row in the tile (alignment slack + head, a multiple of 16, <= kHeadCap), tb = the row's tile
    // offset (exclusive prefix sum); a group = the leading lines whose rows fit the tile
    auto geometry_head = [&](uint64_t q, uint64_t o0, uint64_t o1, uint32_t* nl, uint32_t* span, uint32_t* st_out, uint32_t* tb_out) {
        const uint64_t left = hi_line - q;
        const uint32_t avail = left < (uint64_t)L ? (uint32_t)left : L;  // >= 1
        const uint64_t want = (o1 - o0) + (o0 & 15ull);
        uint32_t st = want >= head_cap<F>::value ? head_cap<F>::value : (uint32_t)((want + 15ull) & ~15ull);
        if (lane >= avail) st = 0u;
        uint32_t total;
        const uint32_t tb = wv::excl_sum(st, &total);
        const unsigned long long fit = __ballot(lane < avail && tb + st <= tile_cap);
        uint32_t cnt = fit == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~fit);
        if (cnt == 0u) cnt = 1u;  // (cannot happen: a row is at most kHeadCap <= tile_cap)
        *nl = cnt;
        const uint32_t last = cnt - 1u;
        *span = (uint32_t)__builtin_amdgcn_readlane((int)tb, (int)last) + (uint32_t)__builtin_amdgcn_readlane((int)st, (int)last);
        *st_out = lane < cnt ? st : 0u;
        *tb_out = tb;
    };
    // row k of the window = the head of line k (a buffer descriptor per row: base = the line's first 16-byte boundary, range = its
    // row's bytes; lanes beyond fetch nothing)
    auto load_window_head = [&](uint64_t o0, uint32_t st, u32x4* v) {
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)o0, k);
            const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(o0 >> 32), k);
            const uint32_t sk = (uint32_t)__builtin_amdgcn_readlane((int)st, k);
            const uint64_t b = ((uint64_t)lo | ((uint64_t)hi << 32)) & ~15ull;
            __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(bytes + b), (short)0, (int)sk, 0x00020000);
            v[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(lane * 16u), 0, FG_STREAM_AUX);
        }
    };
    // the line's last two bytes (the trims and the terminator strip look there)
    auto load_last2 = [&](uint64_t o0, uint64_t o1) -> uint32_t {
        const uint64_t len = o1 - o0;
        uint32_t r = 0;
        if (len >= 1u) r = bytes[o1 - 1u];
        if (len >= 2u) r |= (uint32_t)bytes[o1 - 2u] << 8;
        return r;
    };

    // (the last 8 bytes of the 64-byte pad behind the tile: no tile read reaches them)
    uint32_t* ent_state = reinterpret_cast<uint32_t*>(smem + tile_cap + 56u);
    if (lane < 2u) ent_state[lane] = 0u;
    if (p >= hi_line) return;
    uint64_t o0, o1, a0 = 0;
    uint32_t span, nl, st = 0, tb = 0, last2 = 0;
    load_offsets(p, &o0, &o1);
    u32x4 v[NB];
    if constexpr (HEAD) {
        geometry_head(p, o0, o1, &nl, &span, &st, &tb);
        load_window_head(o0, st, v);
        last2 = load_last2(o0, o1);
    } else {
        geometry(p, o0, o1, &nl, &a0, &span);
        load_window(a0, span, v);
    }
    // The table row of a group is normally stored one iteration LATE (after the next group's stage A,
    // before the prefetch after that is issued): vmcnt retires in order, so stores issued
    // behind the window loads would have to be waited for at the top of every iteration.
    RowOut pend{};
    uint64_t pend_li = 0;
    bool pend_valid = false;
#if defined(FG_ROW64_AB) && FG_ROW64_AB == 2
    // the pending group's records: transposed (on their way to memory) / as each lane packed its own (scalars, not arrays: the arrays stayed in scratch)
    uint4 rq0 = {}, rq1 = {}, rq2 = {}, rq3 = {}, pw0 = {}, pw1 = {}, pw2 = {}, pw3 = {};
    uint64_t rq_p = 0;    // first line of the pending group (wave-uniform)
    uint32_t rq_nl = 0;   // its lines; 0 = nothing pending
    uint64_t pend_p = 0;
    uint32_t pend_nl = 0;
#endif
    uint64_t acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0, iters = 0, tm0 = 0, tm1 = 0, tm2 = 0, tm3 = 0;
    // measurement build only: prof[5] = ablation flags (1 = no table stores, 2 = no stage B,
    // 4 / 8 = format-specific, see the decoders)
    const uint32_t ablate = PROF ? (uint32_t)prof[5] : 0u;
    // the frame terminator the tile's bytes carry BETWEEN lines (never inside one), for formats whose classes care
    const uint32_t term4 = fr.strip == FG_FRAME_LINE ? 0x0A0A0A0Au : fr.strip == FG_FRAME_NUL ? 0u : wv::kNoTerm;

    for (;;) {
        // the next group starts where this one ends: its offsets are requested NOW (they ride behind the window loads and are
        // needed only after stage A, when the next window is issued)
        uint64_t pn = p + nl;
        if (pn >= hi_line) {  // this chunk is done: on to the wave's next one
            if (dyn) {
                if (!tk_pending) draw_ticket();  // (a chunk of one group: drawn and waited for here)
                const uint32_t tk = (uint32_t)__builtin_amdgcn_readfirstlane((int)tk_raw);
                chunk = (uint64_t)(uint32_t)(tk - fr.ticket_base) + G;  // (beyond the last chunk: pn >= n, the wave is done)
                tk_pending = false;
                // the successor's ticket right away: a whole chunk hides its trip.  (Drawn one group before the end, as the first cut
                // did, the tickets of a batch's FIRST round -- every wave starts at the same moment -- arrived in a burst of 1792 on one
                // word and were waited for: 1 M lines 122 us with tickets, 104 without, profiles/r05c_small_ab.log.)
                chunk_range(chunk, kChunkLines, fr.taper0, fr.taper1, fr.taper2, n, &pn, &hi_line);
                if (pn < n) draw_ticket();
            } else {
                chunk += G;
                chunk_range(chunk, kChunkLines, kNoTaper, kNoTaper, kNoTaper, n, &pn, &hi_line);
            }
        }
        const bool more = pn < n;  // wave-uniform
        uint64_t no0 = 0, no1 = 0;
        if (more) load_offsets(pn, &no0, &no1);
        if (PROF) {
            tm0 = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the window has landed
            tm1 = __builtin_amdgcn_s_memtime();
        }
#if defined(FG_ROW64_AB) && FG_ROW64_AB == 2
        rq_nl = 0;
        if (!HEAD && defer_row_store<F>::value) {  // the tile is free: stage B of the last group is done (unconditional: `pend` dies here)
            uint4* rb = reinterpret_cast<uint4*>(smem);
            const uint32_t sw = (lane >> 1) & 3u;  // (16-byte chunks of a record swizzled: lanes two apart would share their banks)
            rb[lane * 4u + (0u ^ sw)] = pw0;
            rb[lane * 4u + (1u ^ sw)] = pw1;
            rb[lane * 4u + (2u ^ sw)] = pw2;
            rb[lane * 4u + (3u ^ sw)] = pw3;
            __syncthreads();
            rq0 = rb[lane];
            rq1 = rb[64u + lane];
            rq2 = rb[128u + lane];
            rq3 = rb[192u + lane];
            __syncthreads();  // (before stage A overwrites the tile)
            rq_p = pend_p;
            rq_nl = pend_nl;
            pend_nl = 0;
            pend_valid = false;
        }
#endif
        // ---- stage A for this group: registers -> LDS, classify on the way ----------------------
        FG_MARK(A);
        const uint32_t nchunk = span >> 4;
        const uint32_t nrow = HEAD ? nl : (nchunk + kWave - 1u) / kWave;  // wave-uniform
        if constexpr (HEAD) {
            // row k = the head of line k: its chunks go to the row's place in the tile
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                if ((uint32_t)k < nl) {  // scalar branch
                    const uint32_t tk = (uint32_t)__builtin_amdgcn_readlane((int)tb, k), sk = (uint32_t)__builtin_amdgcn_readlane((int)st, k);
                    if (lane * 16u < sk) {
                        const uint4 q = make_uint4(v[k][0], v[k][1], v[k][2], v[k][3]);
                        dst[(tk >> 4) + lane] = q;
                        if constexpr (F::kClasses != 0) F::classify_store(q, bm16, (tk >> 4) + lane, bm_stride, term4);
                    }
                }
            }
            // lines beyond the window: HB rows in flight (four; a format whose window is small -- kTailBatch -- takes eight)
            constexpr uint32_t HB = tail_batch<F>::value >= 8 ? 8u : 4u;
            for (uint32_t r0 = NB; r0 < nl; r0 += HB) {
                uint4 w[HB];
                uint32_t tk[HB], sk[HB];
#pragma unroll
                for (uint32_t j = 0; j < HB; ++j) {
                    const uint32_t r = r0 + j < nl ? r0 + j : nl - 1u;
                    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)o0, (int)r);
                    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(o0 >> 32), (int)r);
                    tk[j] = (uint32_t)__builtin_amdgcn_readlane((int)tb, (int)r);
                    sk[j] = (uint32_t)__builtin_amdgcn_readlane((int)st, (int)r);
                    const uint64_t b = ((uint64_t)lo | ((uint64_t)hi << 32)) & ~15ull;
                    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(bytes + b), (short)0, (int)sk[j], 0x00020000);
                    const u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(lane * 16u), 0, FG_STREAM_AUX);
                    w[j] = make_uint4(x[0], x[1], x[2], x[3]);
                }
#pragma unroll
                for (uint32_t j = 0; j < HB; ++j)
                    if (lane * 16u < sk[j]) {  // (a clamped duplicate row writes the same bytes again)
                        dst[(tk[j] >> 4) + lane] = w[j];
                        if constexpr (F::kClasses != 0) F::classify_store(w[j], bm16, (tk[j] >> 4) + lane, bm_stride, term4);
                    }
            }
        } else {
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            if ((uint32_t)k < nrow) {  // scalar branch; lanes past the span store zeros inside the tile
                uint32_t idx = k * kWave + lane;
                uint4 q = make_uint4(v[k][0], v[k][1], v[k][2], v[k][3]);
                dst[idx] = q;
                F::classify_store(q, bm16, idx, bm_stride, idx < nchunk ? term4 : wv::kPastSpan);
            }
        }
        }
        if (!HEAD && nrow > (uint32_t)NB) {  // bytes beyond the register window
            const uint4* __restrict__ src = reinterpret_cast<const uint4*>(bytes + a0);
            // Loads and stores are UNCONDITIONAL (lanes past the end move the last chunk once more: same bytes, same
            // address): with `if (idx < nchunk)` the compiler kept w[] in scratch memory and waited for each load
            // before issuing the next -- serialised HBM round trips instead of four loads in flight.
            const uint32_t last = nchunk - 1u;
            // (TB loads in flight per lane: a format whose tile is mostly staged HERE -- a 2 KiB window under an 18 KiB tile -- pays
            //  the load latency once per TB KiB, and stage B's registers are dead at this point)
            constexpr int TB = tail_batch<F>::value;
            for (uint32_t c0 = NB * kWave; c0 < nchunk; c0 += kWave * TB) {
                uint4 w[TB];  // (the window registers are dead here)
#pragma unroll
                for (int k = 0; k < TB; ++k) {
                    const uint32_t idx = c0 + k * kWave + lane;
                    w[k] = stream_load(src + (idx < last ? idx : last));
                }
#pragma unroll
                for (int k = 0; k < TB; ++k) {
                    if (k == 0 || c0 + k * kWave < nchunk) {  // (scalar: whole rows past the end are skipped)
                        const uint32_t idx = c0 + k * kWave + lane, ci = idx < last ? idx : last;
                        dst[ci] = w[k];
                        F::classify_store(w[k], bm16, ci, bm_stride, term4);
                    }
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the old window dead before the new one is loaded
        FG_MARK(S);
        if (PROF) {
            __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): LDS writes retired
            tm2 = __builtin_amdgcn_s_memtime();
        }
#if defined(FG_ROW64_AB) && FG_ROW64_AB == 2
        if (rq_nl != 0u && !(ablate & 1u)) {
            uint4* g = reinterpret_cast<uint4*>(t.meta) + rq_p * 4u;
            if (lane < rq_nl * 4u) g[lane] = rq0;
            if (64u + lane < rq_nl * 4u) g[64u + lane] = rq1;
            if (128u + lane < rq_nl * 4u) g[128u + lane] = rq2;
            if (192u + lane < rq_nl * 4u) g[192u + lane] = rq3;
        }
#elif defined(FG_ROW64_AB)
        if (pend_valid && !(ablate & 1u)) store_row64_lane(t, pend_li, pend);
#else
        if (pend_valid && !(ablate & 1u)) store_row(t, pend_li, pend);
#endif
        // ---- prefetch: the next group's geometry from its offsets, then its bytes into the register window ----
        uint64_t pa0 = 0;
        uint32_t pspan = 0, pnl = 0, pst = 0, ptb = 0, plast2 = 0;
        if (more) {
            if constexpr (HEAD) {
                geometry_head(pn, no0, no1, &pnl, &pspan, &pst, &ptb);
                load_window_head(no0, pst, v);
                plast2 = load_last2(no0, no1);
            } else {
                geometry(pn, no0, no1, &pnl, &pa0, &pspan);
                load_window(pa0, pspan, v);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        FG_MARK(B);
        if (PROF) tm3 = __builtin_amdgcn_s_memtime();
        __syncthreads();  // single-wave workgroup: orders the LDS writes before stage B's reads
        // ---- stage B for this group ---------------------------------------------------------
        if (!(ablate & 2u)) {
            const uint64_t li = p + lane;
            const bool valid = lane < nl;
            // terminator stripping (BufRead::lines / split(0) semantics, see fg_frame.hip)
            uint64_t e1 = o1;
            uint32_t l2 = last2;  // (HEAD) the last two bytes of the line AS DECODED, i.e. behind its terminator
            if (fr.strip != FG_FRAME_NONE && valid && e1 > o0) {
                auto byte_at = [&](uint64_t q) -> uint32_t {
                    if constexpr (HEAD) return (uint32_t)bytes[q];
                    return (q - a0) < (uint64_t)span ? (uint32_t)smem[q - a0] : (uint32_t)bytes[q];
                };
                const uint32_t b1 = HEAD ? (last2 & 0xFFu) : byte_at(e1 - 1);
                if (fr.strip == FG_FRAME_LINE) {
                    if (b1 == '\n') {
                        --e1;
                        if (e1 > o0 && (HEAD ? ((last2 >> 8) & 0xFFu) : byte_at(e1 - 1)) == '\r') --e1;
                    }
                } else if (b1 == 0u) {
                    --e1;
                }
                if (HEAD && e1 != o1) {  // (rare in HEAD mode -- long lines come framed by offsets -- so: two plain loads)
                    l2 = 0u;
                    if (e1 - o0 >= 1u) l2 = byte_at(e1 - 1);
                    if (e1 - o0 >= 2u) l2 |= byte_at(e1 - 2) << 8;
                }
            }
            // Every line of the group lies inside the tile by construction -- except a single line longer than the whole tile,
            // which is a group of its own and is parsed straight from global memory (the decoders look at o1 - a0 <= span).
            GroupCtx c{bytes, smem, bm16, o0, e1, a0, span, valid, li, (ablate & 8u) ? nullptr : stash, ablate, ent_state, PROF ? prof + 6 : nullptr};
            c.tile_cap = tile_cap;
            if constexpr (HEAD) {
                const uint32_t al = (uint32_t)(o0 & 15ull), len = (uint32_t)(e1 - o0);
                c.tbase = tb + al;
                c.tlen = st == 0u ? 0u : (st - al < len ? st - al : len);
                c.last2 = l2;
            }
            pend = fmt.decode(c, t);
            if (fr.line_bad && valid && fr.line_bad[li]) {  // "Invalid UTF-8 input": the frame never reaches decode()
                pend.meta = FG_ST_BAD_UTF8 | (0xFFu << 8) | (0xFFu << 16);
                pend.ts = 0.0;
#pragma unroll
                for (int k = 0; k < 6; ++k) pend.span[k] = fg_span{0, FG_NONE};
                pend.count = 0;
            }
            pend_li = li;
            pend_valid = valid;
#if defined(FG_ROW64_AB) && FG_ROW64_AB == 2
            pend_p = p;
            pend_nl = nl;
            if (!HEAD && defer_row_store<F>::value) {
                uint4 w[4];
                pack_row64(pend, w);
                pw0 = w[0], pw1 = w[1], pw2 = w[2], pw3 = w[3];
            }
#endif
            if (!defer_row_store<F>::value) {  // (a format whose stage A waits for its own loads anyway: nothing to hide behind)
#if defined(FG_ROW64_AB)
                if (pend_valid && !(ablate & 1u)) store_row64_lane(t, pend_li, pend);
#else
                if (pend_valid && !(ablate & 1u)) store_row(t, pend_li, pend);
#endif
                pend_valid = false;
            }
        }
        if (PROF) {
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xC07F);
            uint64_t tm4 = __builtin_amdgcn_s_memtime();
            acc0 += tm1 - tm0;
            acc1 += tm2 - tm1;
            acc2 += tm3 - tm2;
            acc3 += tm4 - tm3;
            iters += 1;
        }
        FG_MARK(Z);
        if (!more) break;
        __syncthreads();  // stage B's LDS reads are done before the next tile overwrites them
        p = pn;
        nl = pnl;
        o0 = no0;
        o1 = no1;
        a0 = pa0;
        span = pspan;
        if constexpr (HEAD) {
            st = pst;
            tb = ptb;
            last2 = plast2;
        }
    }
#if defined(FG_ROW64_AB) && FG_ROW64_AB == 2
    if (!HEAD && defer_row_store<F>::value) {
        if (pend_valid && !(ablate & 1u)) {
            uint4* g = reinterpret_cast<uint4*>(t.meta) + pend_li * 4u;
            g[0] = pw0, g[1] = pw1, g[2] = pw2, g[3] = pw3;
        }
    } else if (pend_valid && !(ablate & 1u)) {
        store_row64_lane(t, pend_li, pend);
    }
#elif defined(FG_ROW64_AB)
    if (pend_valid && !(ablate & 1u)) store_row64_lane(t, pend_li, pend);
#else
    if (pend_valid && !(ablate & 1u)) store_row(t, pend_li, pend);
#endif
    if (PROF && lane == 0) {
        atomicAdd(&prof[0], (unsigned long long)acc0);
        atomicAdd(&prof[1], (unsigned long long)acc1);
        atomicAdd(&prof[2], (unsigned long long)acc2);
        atomicAdd(&prof[3], (unsigned long long)acc3);
        atomicAdd(&prof[4], (unsigned long long)iters);
    }
}

}  // namespace fg

// ---------------------------------------------------------------------------------------------
// Host side: launch geometry shared by the three launchers
// ---------------------------------------------------------------------------------------------
#include <cstdio>
#include <cstdlib>

namespace fg {
constexpr int kWindowKiB = 20;  // register prefetch window per wave (NB): 80 VGPRs
// The LTSV / GELF tokenisers are compute-bound (hundreds of instructions per line) and register
// hungry: they keep only a token window (the rest of a group is staged by the plain tail loop,
// ~2 us of exposed latency against a group time of tens of us) and get the registers instead.
constexpr int kComputeBoundWindow = 2;

struct LaunchPlan {
    uint32_t L = 64;      // lines per group
    uint32_t tile = 0;    // LDS tile bytes (multiple of 1024)
    uint32_t lds = 0;     // dynamic LDS bytes per workgroup = tile + 64 + bitmap + extra
    uint64_t groups = 0;  // an estimate (the waves cut their ranges into groups themselves)
    uint64_t chunk = 256; // lines a wave takes at a time (the first by block index, the rest by ticket)
    uint64_t chunks = 0;  // ceil(n / chunk) >= blocks
    bool tickets = false; // chunks beyond a wave's first are drawn from the launch's ticket counter (else: round-robin)
    uint32_t taper[3] = {kNoTaper, kNoTaper, kNoTaper};  // ticket launches: where the chunks shrink (fg_plan_policy.hpp)
    uint32_t blocks = 0;  // persistent grid
};

// The launch's ticket counter into the kernel's arguments; the host's copy of the counter moves on by what the launch will draw
// (every wave draws until its first ticket beyond the last chunk: chunks - blocks good ones + blocks bad ones = chunks).
inline void take_tickets(FrameArgs* fr, TicketSlot* tk, const LaunchPlan& p) {
    if (!tk || !tk->d_word || !tk->h_val || p.chunks + p.blocks >= 0xFFFFFFFFull || p.blocks > p.chunks) return;  // (static round-robin)
    if (!p.tickets) return;  // (equal shares: nothing to draw)
    fr->ticket = tk->d_word;
    fr->ticket_base = *tk->h_val;
    fr->taper0 = p.taper[0], fr->taper1 = p.taper[1], fr->taper2 = p.taper[2];
    *tk->h_val += (uint32_t)p.chunks;
}

// Lines per group L and the LDS tile: the largest power of two L <= 64 whose average group
// (+6.25 % + 256 B) fits the register prefetch window, so that a whole group is prefetched.
// Lines longer than the window get L = 1 and a tile of up to `max_tile` (the part beyond the
// window is staged by the tail loop); anything that still does not fit is parsed from global
// memory.  fg_launch_opts (fg_set_launch_opts: tuning, parity sweeps) overrides tile / lines per group / waves per CU; the
// library itself reads no environment variable.
// What a format says about its own launch, by NAME (twelve positional arguments ending `64, 1, nullptr, nullptr, 0u, 0u, 20u, 0u`
// were one swapped literal away from a silent performance regression): PlanFormat().classes(2).extra_tile(f).tile(12288).chunk(128) ...
struct PlanFormat {
    uint32_t max_lines = 64;        // lines a group holds at most (the wave width, or the format's cap)
    uint32_t n_classes = 1;         // stage-A class bitmaps kept in LDS
    uint32_t (*extra_for)(uint32_t tile, uint32_t lines) = nullptr;  // LDS that grows with the tile and the lines per group (per-item arrays)
    uint32_t (*extra_tile)(uint32_t tile) = nullptr;                 // LDS that grows with the tile only
    uint32_t default_tile = 0;      // a cap on the tile the format gets by default (its LDS beyond the tile grows with it)
    uint32_t default_chunk = 0;     // lines a wave takes at a time under ticket dispatch (0: 256 / 512 by the lines a tile holds)
    uint32_t ticket_from = 2;       // chunks of that size per wave from which chunks are drawn by ticket
    uint32_t taper = 1;             // levels of the taper at the end of a ticket launch
    PlanFormat& lines(uint32_t v) { max_lines = v; return *this; }
    PlanFormat& classes(uint32_t v) { n_classes = v; return *this; }
    PlanFormat& lds_for(uint32_t (*f)(uint32_t, uint32_t)) { extra_for = f; return *this; }
    PlanFormat& lds_tile(uint32_t (*f)(uint32_t)) { extra_tile = f; return *this; }
    PlanFormat& tile(uint32_t v) { default_tile = v; return *this; }
    PlanFormat& chunk(uint32_t v) { default_chunk = v; return *this; }
    PlanFormat& tickets_from(uint32_t v) { ticket_from = v; return *this; }
    PlanFormat& taper_levels(uint32_t v) { taper = v; return *this; }
};
template <class K>
inline int plan_launch(K kernel, uint64_t n, uint64_t avg_len, uint32_t extra_lds, uint32_t max_tile, uint32_t stash_blocks, LaunchPlan* p,
                       const fg_launch_opts& lo, const PlanFormat& pf = PlanFormat()) {
    const uint32_t max_lines = pf.max_lines, n_classes = pf.n_classes, default_tile = pf.default_tile, default_chunk = pf.default_chunk;
    const uint32_t ticket_from = pf.ticket_from, taper = pf.taper;
    uint32_t (*const extra_for)(uint32_t, uint32_t) = pf.extra_for;
    uint32_t (*const extra_tile)(uint32_t) = pf.extra_tile;
    const uint64_t window = (uint64_t)kWindowKiB * 1024u;
    // (+6.25 % + 256 B over the average group: a few sigma for the corpora at hand; a longer group just takes
    //  another pass over a restaged tile, while every KiB of LDS saved is occupancy)
    auto tile_for = [&](uint32_t l) { return (((uint64_t)l * avg_len * 17u / 16u + 256u) + 1023u) / 1024u * 1024u; };
    auto clamp = [&](uint64_t v) { return (uint32_t)(v < 4096u ? 4096u : v > max_tile ? max_tile : v); };
    uint32_t L = max_lines;
    // (up to two 1-KiB rows beyond the window are tolerated: the plain tail loop stages them; measured on
    //  the 554-byte structured-data corpus L = 32 at 6 waves/CU beats L = 16 at 8 by 30 %)
    while (L > 1 && tile_for(L) > window + 2048u) L >>= 1;
    uint32_t tile = clamp(tile_for(L));
    if (lo.lines_per_group >= 1 && lo.lines_per_group <= max_lines) {
        L = lo.lines_per_group;
        tile = clamp(tile_for(L));
    }
    // (default_tile: a format whose LDS beyond the tile grows with it caps the tile it gets by default)
    if (default_tile && tile > default_tile && !(lo.lines_per_group >= 1 && lo.lines_per_group <= max_lines)) tile = default_tile;
    if (lo.tile_cap >= 1024 && lo.tile_cap <= max_tile) tile = (lo.tile_cap + 1023u) / 1024u * 1024u;
    // Groups are cut BY BYTES (persistent_loop): L is only the cap on the lines of a group.  A format without per-line LDS arrays
    // takes the full wave width -- however long the lines, a group then holds as many as fit the tile; the tile of long lines
    // (fewer than 16 average lines in the window) is raised to what eight waves per CU leave each other anyway.
    if (!extra_for && !(lo.lines_per_group >= 1 && lo.lines_per_group <= max_lines)) {
        if (L <= 16u && !(lo.tile_cap >= 1024 && lo.tile_cap <= max_tile) && tile < 18432u && max_tile >= 18432u && !default_tile) tile = 18432u;
        L = max_lines;
    }
    p->L = L;
    p->tile = tile;
    // (extra_for: LDS a format needs as a function of the tile and the lines per group, e.g. per-item arrays)
    p->lds = tile + 64u + (tile / 16u + 16u) * 2u * (n_classes ? n_classes : 1u) + extra_lds + (extra_for ? extra_for(tile, L) : 0u) +
             (extra_tile ? extra_tile(tile) : 0u);
    {   // an estimate (the waves cut their ranges themselves): by lines and by bytes
        const uint64_t by_lines = (n + L - 1) / L, by_bytes = (n * avg_len + tile - 1) / tile;
        p->groups = by_lines > by_bytes ? by_lines : by_bytes;
    }
    int dev = 0, cus = 0;  // (per call: a process may drive several devices)
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1)
        return -1;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kWave, p->lds) != hipSuccess || per_cu < 1) per_cu = 1;
    if (lo.waves_per_cu > 0 && per_cu > (int)lo.waves_per_cu) per_cu = (int)lo.waves_per_cu;
    uint64_t blocks = (uint64_t)per_cu * (uint64_t)cus;
    if (blocks > p->groups) blocks = p->groups;
    if (stash_blocks && blocks > stash_blocks) blocks = stash_blocks;
    // 256 lines per chunk (four groups of 64 short lines: the HBM-bound configuration keeps its sweep tight) -- 512 when a group
    // holds fewer than L average lines, so that the short group at the end of every chunk stays a few per cent (same-box sweep,
    // tools/sweep.py, 4 M lines: cfg4 1058 / 1150 / 974 M lines/s at 256 / 512 / 1024 -- the grid's 1792 waves need a few rounds
    // of chunks each to finish together)
    // (default_chunk: a format's own choice -- fg_rfc5424.hip, fg_gelf.hip, fg_ltsv.hip: each from a same-box sweep under ticket dispatch)
    const uint64_t full = default_chunk ? default_chunk : (avg_len ? (uint64_t)p->tile / avg_len : p->L) >= p->L ? 256u : 512u;
    // lines an average group holds (groups are cut by bytes): the unit chunks are made of
    const uint64_t g = avg_len ? ((uint64_t)p->tile * 16u) / (avg_len * 17u) : p->L;
    // how the batch is cut into chunks, and whether chunks are drawn by ticket: pure arithmetic, fg_plan_policy.hpp (CPU-tested)
    const ChunkPlan cp = plan_chunks(n, blocks, p->L, g, full, ticket_from, lo, taper);
    p->chunk = cp.chunk;
    p->chunks = cp.chunks;
    p->tickets = cp.tickets;
    for (int j = 0; j < 3; ++j) p->taper[j] = cp.taper[j];
    p->blocks = cp.blocks;
    return 0;
}

// The measurement build (-DFG_PROF_BUILD, `FG_BUILD_PROF=1 python -m flowgger_amd.build` -> libfg_hip_prof.so) also compiles the
// PROF = true instantiation of each kernel; FG_PROF=1 in the environment then runs it synchronously and prints the per-phase split
// (FG_ABLATE: ablation flags).  The product library has neither the instantiations nor any read of the environment.
#if defined(FG_PROF_BUILD)
inline bool prof_requested() { return getenv("FG_PROF") != nullptr; }
#else
constexpr bool prof_requested() { return false; }
#endif
struct ProfRun {
    unsigned long long* d = nullptr;
    unsigned long long h[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // [6..15]: format-specific phase clocks (GroupCtx::phase)
    bool begin(hipStream_t stream) {
#if defined(FG_PROF_BUILD)
        if (const char* e = getenv("FG_ABLATE")) h[5] = (unsigned long long)atoi(e);
#endif
        if (hipMalloc((void**)&d, sizeof(h)) != hipSuccess) return false;
        (void)hipMemcpyAsync(d, h, sizeof(h), hipMemcpyHostToDevice, stream);
        return true;
    }
    void end(hipStream_t stream, const char* name, const LaunchPlan& p) {
        (void)hipStreamSynchronize(stream);
        (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        (void)hipFree(d);
        const double it = h[4] ? (double)h[4] : 1.0;
        fprintf(stderr,
                "[fg prof] %s: grid %u x64, L %u, tile %u, iters/wave %.1f | cycles per iteration: wait %.0f, stageA %.0f, "
                "stores+prefetch-issue %.0f, stageB %.0f\n",
                name, p.blocks, p.L, p.tile, it / (double)(p.blocks ? p.blocks : 1), h[0] / it, h[1] / it, h[2] / it, h[3] / it);
        bool any_phase = false;
        for (int k = 6; k < 16; ++k) any_phase = any_phase || h[k] != 0;
        if (any_phase) {
            fprintf(stderr, "[fg prof] %s stage-B phases (cycles per iteration):", name);
            for (int k = 6; k < 16; ++k) fprintf(stderr, " %.0f", h[k] / it);
            fprintf(stderr, "\n");
        }
    }
};
}  // namespace fg
