// fg_device.hpp -- device-side building blocks shared by the gfx950 decode kernels.
//
// Everything here is integer / byte work executed lane-per-line out of an LDS-staged tile of the
// packed line buffer (the tile is brought in with coalesced 16-byte-per-lane loads).  The only
// floating point is the final timestamp conversion, which must reproduce
//   PreciseTimestamp::from_offset_datetime: unix_timestamp_nanos() as f64 / 1e9
//   (reference: src/flowgger/utils/mod.rs:23-28)
// bit for bit: i128 -> f64 round-to-nearest-even, then one IEEE-754 division.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fg_hip.h"
#include "fg_tables_view.hpp"
#include "fg_timeconv.hpp"

namespace fg {

constexpr int kWave = 64;

// ---------------------------------------------------------------------------------------------
// Byte readers.  A reader serves bytes of ONE line by index (0 .. len) and keeps the last
// aligned dword in a register, so a sequential walk costs one LDS (or global) access per 4 bytes.
// ---------------------------------------------------------------------------------------------
struct LdsReader {
    const uint32_t* words;  // LDS tile viewed as dwords
    uint32_t base;          // byte offset of the line inside the tile
    uint32_t cur_idx = 0xFFFFFFFFu;
    uint32_t cur = 0;
    __device__ __forceinline__ LdsReader(const uint32_t* w, uint32_t b) : words(w), base(b) {}
    __device__ __forceinline__ uint32_t byte(uint32_t i) {
        uint32_t a = base + i;
        uint32_t w = a >> 2;
        if (w != cur_idx) {
            cur = words[w];
            cur_idx = w;
        }
        return __builtin_amdgcn_ubfe(cur, (a & 3u) * 8u, 8u);
    }
    // 4 bytes starting at byte i (little-endian); only the low nb are meaningful.  The tile is padded: the dword
    // behind the last byte is always readable.
    __device__ __forceinline__ uint32_t load4(uint32_t i, uint32_t) {
        const uint32_t a = base + i;
        const uint32_t w = a >> 2;
        if (w != cur_idx) {
            cur = words[w];
            cur_idx = w;
        }
        const uint32_t lo = cur;
        if ((a & 3u) == 0) return lo;
        cur = words[w + 1u];
        cur_idx = w + 1u;
        return __builtin_amdgcn_alignbyte(cur, lo, a & 3u);
    }
    // 16 bytes starting at byte i, all wanted: five independent dword reads in flight (ONE LDS round trip instead of
    // four dependent ones); the fifth is only looked at when the start is not dword aligned
    __device__ __forceinline__ void load16(uint32_t i, uint32_t* q) {
        const uint32_t a = base + i, w = a >> 2, sh = a & 3u;
        const uint32_t d0 = words[w], d1 = words[w + 1u], d2 = words[w + 2u], d3 = words[w + 3u], d4 = words[w + (sh ? 4u : 3u)];
        q[0] = __builtin_amdgcn_alignbyte(d1, d0, sh);
        q[1] = __builtin_amdgcn_alignbyte(d2, d1, sh);
        q[2] = __builtin_amdgcn_alignbyte(d3, d2, sh);
        q[3] = __builtin_amdgcn_alignbyte(d4, d3, sh);
    }
    // 16 bytes starting at byte i of which only the first nb (1 .. 15) are wanted; the others come back unspecified.  The tile is
    // followed by 16 bytes of padding and the launch's configuration mirror: the reads stay inside the workgroup's LDS.
    __device__ __forceinline__ void load16p(uint32_t i, uint32_t, uint32_t* q) { load16(i, q); }
};
struct GlobalReader {
    // Round 5: the reader keeps the ALIGNED 16 bytes around its position (one dwordx4 load) instead of one dword: a byte-wise walk
    // through global memory -- a line longer than the tile, a structured-data block that runs past its staged head -- pays one trip
    // to memory per 16 bytes, not per 4.  (That walk is the floor of a small batch's kernel time: a 1 KiB walk by ONE lane was ~0.3 ms
    // at ~1 us per dependent load, whatever the other 130 000 lanes of the grid did.)  The packed buffer is 16-byte aligned and
    // readable up to its size rounded up to 16 (include/fg_hip.h), so the aligned chunk around any byte of a line is readable.
    const uint32_t* words;  // packed buffer viewed as dwords (base is 16-byte aligned)
    uint64_t base;          // byte offset of the line inside the packed buffer
    uint64_t cur_idx = ~0ull;  // index of the 16-byte chunk held
    uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    __device__ __forceinline__ GlobalReader(const uint32_t* w, uint64_t b) : words(w), base(b) {}
    __device__ __forceinline__ void fill(uint64_t chunk) {
        // (four dwords of one aligned chunk: the compiler makes it one global_load_dwordx4)
        const uint32_t* q = static_cast<const uint32_t*>(__builtin_assume_aligned(words + chunk * 4u, 16));
        c0 = q[0];
        c1 = q[1];
        c2 = q[2];
        c3 = q[3];
        cur_idx = chunk;
    }
    // dword k (0..3) of the chunk held -- by selects: a dynamically indexed register array would live in scratch memory
    __device__ __forceinline__ uint32_t dw(uint32_t k) const { return k == 0u ? c0 : k == 1u ? c1 : k == 2u ? c2 : c3; }
    __device__ __forceinline__ uint32_t byte(uint32_t i) {
        const uint64_t a = base + i;
        if ((a >> 4) != cur_idx) fill(a >> 4);
        return __builtin_amdgcn_ubfe(dw(((uint32_t)a >> 2) & 3u), ((uint32_t)a & 3u) * 8u, 8u);
    }
    // 4 bytes starting at byte i; the chunk behind is only touched when the nb wanted bytes reach into it (the
    // packed buffer is readable up to nbytes rounded up to 16, not beyond)
    __device__ __forceinline__ uint32_t load4(uint32_t i, uint32_t nb) {
        const uint64_t a = base + i;
        if ((a >> 4) != cur_idx) fill(a >> 4);
        const uint32_t k = ((uint32_t)a >> 2) & 3u, sh = (uint32_t)a & 3u;
        const uint32_t lo = dw(k);
        if (sh + nb <= 4u) return lo >> (8u * sh);
        uint32_t hi;
        if (k < 3u) {
            hi = dw(k + 1u);
        } else {
            fill((a >> 4) + 1u);
            hi = c0;
        }
        return __builtin_amdgcn_alignbyte(hi, lo, sh);
    }
    // 16 wanted bytes starting at byte i (see LdsReader::load16); never reads a dword without wanted bytes
    __device__ __forceinline__ void load16(uint32_t i, uint32_t* q) {
        const uint64_t a = base + i, w = a >> 2;
        const uint32_t sh = (uint32_t)a & 3u;
        const uint32_t d0 = words[w], d1 = words[w + 1u], d2 = words[w + 2u], d3 = words[w + 3u], d4 = words[w + (sh ? 4u : 3u)];
        q[0] = __builtin_amdgcn_alignbyte(d1, d0, sh);
        q[1] = __builtin_amdgcn_alignbyte(d2, d1, sh);
        q[2] = __builtin_amdgcn_alignbyte(d3, d2, sh);
        q[3] = __builtin_amdgcn_alignbyte(d4, d3, sh);
    }
    // the first nb (1 .. 15) of 16 bytes starting at byte i, dword by dword: never reads a chunk without wanted bytes (this reader
    // serves the rare group that does not fit its tile)
    __device__ __forceinline__ void load16p(uint32_t i, uint32_t nb, uint32_t* q) {
#pragma unroll
        for (uint32_t j = 0; j < 4u; ++j) q[j] = nb > 4u * j ? load4(i + 4u * j, nb - 4u * j < 4u ? nb - 4u * j : 4u) : 0u;
    }
};

// ---------------------------------------------------------------------------------------------
// Unicode White_Space trimming (Rust str::trim / trim_end; rfc5424_decoder.rs:46,167).
// White_Space = U+0009..000D, 0020, 0085, 00A0, 1680, 2000..200A, 2028, 2029, 202F, 205F, 3000.
// UTF-8: C2 85 | C2 A0 | E1 9A 80 | E2 80 (80..8A|A8|A9|AF) | E2 81 9F | E3 80 80.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool ascii_ws(uint32_t c) { return c == 32u || (c - 9u) <= 4u; }
__device__ __forceinline__ bool ws3(uint32_t b0, uint32_t b1, uint32_t b2) {
    if (b0 == 0xE2u) {
        if (b1 == 0x80u) return (b2 - 0x80u) <= 0x0Au || b2 == 0xA8u || b2 == 0xA9u || b2 == 0xAFu;
        return b1 == 0x81u && b2 == 0x9Fu;
    }
    if (b0 == 0xE1u) return b1 == 0x9Au && b2 == 0x80u;
    if (b0 == 0xE3u) return b1 == 0x80u && b2 == 0x80u;
    return false;
}
// returns the new end (exclusive) of [s, e) after trimming trailing whitespace
template <class R>
__device__ __forceinline__ uint32_t trim_end(R& rd, uint32_t s, uint32_t e) {
    while (e > s) {
        uint32_t c = rd.byte(e - 1);
        if (c < 0x80u) {
            if (!ascii_ws(c)) break;
            e -= 1;
        } else if (e - s >= 2 && (c == 0x85u || c == 0xA0u) && rd.byte(e - 2) == 0xC2u) {
            e -= 2;
        } else if (e - s >= 3 && ws3(rd.byte(e - 3), rd.byte(e - 2), c)) {
            e -= 3;
        } else {
            break;
        }
    }
    return e;
}
template <class R>
__device__ __forceinline__ uint32_t trim_start(R& rd, uint32_t s, uint32_t e) {
    while (s < e) {
        uint32_t c = rd.byte(s);
        if (c < 0x80u) {
            if (!ascii_ws(c)) break;
            s += 1;
        } else if (c == 0xC2u && e - s >= 2) {
            uint32_t d = rd.byte(s + 1);
            if (d != 0x85u && d != 0xA0u) break;
            s += 2;
        } else if (e - s >= 3 && ws3(c, rd.byte(s + 1), rd.byte(s + 2))) {
            s += 3;
        } else {
            break;
        }
    }
    return s;
}

// Calendar + timestamp arithmetic: fg_timeconv.hpp (host-testable).

// n ASCII digits at rd[q..q+n) (all inside [0,len)); advances q.
template <class R>
__device__ __forceinline__ bool take_digits(R& rd, uint32_t& q, uint32_t len, int n, int* out) {
    if (q + (uint32_t)n > len) return false;
    int v = 0;
    for (int k = 0; k < n; ++k) {
        uint32_t d = rd.byte(q + k) - '0';
        if (d > 9u) return false;
        v = v * 10 + (int)d;
    }
    q += n;
    *out = v;
    return true;
}
template <class R>
__device__ __forceinline__ bool take_subsecond(R& rd, uint32_t& q, uint32_t len, uint32_t* nano) {
    if (q >= len) return false;
    uint32_t d = rd.byte(q) - '0';
    if (d > 9u) return false;
    uint32_t v = d * 100000000u, mult = 10000000u;
    ++q;
    while (q < len) {
        d = rd.byte(q) - '0';
        if (d > 9u) break;
        v += d * mult;
        mult /= 10u;
        ++q;
    }
    *nano = v;
    return true;
}
// time::OffsetDateTime::parse(s, &Rfc3339) over rd[q..end): YYYY-MM-DD[Tt]HH:MM:SS[.d+]([Zz]|[+-]HH:MM),
// whole [q,end) consumed.  (rfc5424_decoder.rs:94-99, ltsv_decoder.rs:224-229)
template <class R>
__device__ __forceinline__ bool parse_rfc3339(R& rd, uint32_t q, uint32_t end, double* out) {
    DateTimeParts p;
    if (!take_digits(rd, q, end, 4, &p.year)) return false;
    if (q >= end || rd.byte(q) != '-') return false;
    ++q;
    if (!take_digits(rd, q, end, 2, &p.month)) return false;
    if (q >= end || rd.byte(q) != '-') return false;
    ++q;
    if (!take_digits(rd, q, end, 2, &p.day)) return false;
    if (q >= end || (rd.byte(q) | 0x20u) != 't') return false;
    ++q;
    if (!take_digits(rd, q, end, 2, &p.hour)) return false;
    if (q >= end || rd.byte(q) != ':') return false;
    ++q;
    if (!take_digits(rd, q, end, 2, &p.minute)) return false;
    if (q >= end || rd.byte(q) != ':') return false;
    ++q;
    if (!take_digits(rd, q, end, 2, &p.second)) return false;
    p.nano = 0;
    if (q < end && rd.byte(q) == '.') {
        ++q;
        if (!take_subsecond(rd, q, end, &p.nano)) return false;
    }
    p.off_sign = 1;
    p.off_h = 0;
    p.off_m = 0;
    if (q >= end) return false;
    uint32_t c = rd.byte(q);
    if ((c | 0x20u) == 'z') {
        ++q;
    } else {
        if (c != '+' && c != '-') return false;
        p.off_sign = c == '-' ? -1 : 1;
        ++q;
        if (!take_digits(rd, q, end, 2, &p.off_h)) return false;
        if (q >= end || rd.byte(q) != ':') return false;
        ++q;
        if (!take_digits(rd, q, end, 2, &p.off_m)) return false;
    }
    if (q != end) return false;
    return datetime_to_unix(p, true, out);
}

// Stream [a0, a0+span) of the packed buffer into the wave's LDS tile: 16 B per lane, 1 KiB per
// wave-instruction, 8 loads in flight per lane (a0 and span are multiples of 16).
// Loads AND stores are unconditional (lanes past the end move the last chunk once more, same bytes to the same
// address): with `if (idx < nchunk)` around either, the compiler kept v[] in scratch memory or sank each load into
// its store's branch, and waited for every load before issuing the next one -- eight serialised HBM round trips per
// 8 KiB instead of eight loads in flight (round-1 finding).
// K = 16-byte loads in flight per lane (K KiB per round trip; 4 K VGPRs that are dead again afterwards).
template <int K = 8>
__device__ __forceinline__ void stage_tile(const uint8_t* __restrict__ bytes, uint64_t a0, uint32_t span, uint8_t* smem) {
    const uint4* __restrict__ src = reinterpret_cast<const uint4*>(bytes + a0);
    uint4* dst = reinterpret_cast<uint4*>(smem);
    const uint32_t nchunk = span >> 4;
    if (nchunk == 0) return;
    const uint32_t lane = threadIdx.x;
    const uint32_t last = nchunk - 1u;
    for (uint32_t c0 = 0; c0 < nchunk; c0 += kWave * K) {
        uint4 v[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint32_t idx = c0 + k * kWave + lane;
            v[k] = src[idx < last ? idx : last];
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint32_t idx = c0 + k * kWave + lane;
            dst[idx < last ? idx : last] = v[k];
        }
    }
}

// stage_tile plus a small second block (at most 4 KiB = four 16-byte loads per lane, e.g. a configuration mirror)
// whose loads ride in the same round trip: tile loads, rider loads, tile stores, rider stores -- straight-line code for
// the first K KiB so that the compiler has no branch to sink the rider's loads behind.
template <int K>
__device__ __forceinline__ void stage_tile_rider(const uint8_t* __restrict__ bytes, uint64_t a0, uint32_t span, uint8_t* smem,
                                                 const uint4* __restrict__ rider_src, uint32_t rider_chunks, uint4* rider_dst) {
    const uint4* __restrict__ src = reinterpret_cast<const uint4*>(bytes + a0);
    uint4* dst = reinterpret_cast<uint4*>(smem);
    const uint32_t nchunk = span >> 4;
    const uint32_t lane = threadIdx.x;
    const uint32_t last = nchunk ? nchunk - 1u : 0u;   // (span == 0: the tile's 16 spare bytes take chunk 0 of the buffer's padding)
    const uint32_t rlast = rider_chunks ? rider_chunks - 1u : 0u;
    uint4 v[K], r[4];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const uint32_t idx = k * kWave + lane;
        v[k] = src[idx < last ? idx : last];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t idx = k * kWave + lane;
        r[k] = rider_src[idx < rlast ? idx : rlast];
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const uint32_t idx = k * kWave + lane;
        dst[idx < last ? idx : last] = v[k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t idx = k * kWave + lane;
        rider_dst[idx < rlast ? idx : rlast] = r[k];
    }
    for (uint32_t c0 = kWave * K; c0 < nchunk; c0 += kWave * K) {  // rare: a tile beyond K KiB
        uint4 w[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint32_t idx = c0 + k * kWave + lane;
            w[k] = src[idx < last ? idx : last];
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint32_t idx = c0 + k * kWave + lane;
            dst[idx < last ? idx : last] = w[k];
        }
    }
}

// wave-wide exclusive prefix sum of a 32-bit value; *total receives the wave sum.
__device__ __forceinline__ uint32_t wave_exclusive_sum(uint32_t v, uint32_t* total) {
    uint32_t lane = __lane_id();
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        uint32_t t = __shfl_up(inc, d, kWave);
        if (lane >= (uint32_t)d) inc += t;
    }
    *total = __shfl(inc, kWave - 1, kWave);
    return inc - v;
}

}  // namespace fg
