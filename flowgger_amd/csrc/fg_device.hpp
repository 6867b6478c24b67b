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

namespace fg {

constexpr int kWave = 64;

// Device view of fg_tables (same arrays, device pointers).
struct DevTables {
    uint64_t n;
    uint64_t ent_cap;
    uint32_t* meta;
    double* ts;
    fg_span* span[6];  // hostname, appname, procid, msgid, msg, full_msg
    uint32_t* ent_first;
    uint32_t* ent_count;
    fg_span* ent_name;
    uint64_t* ent_val;
    uint8_t* ent_type;
    uint8_t* ent_flags;
    unsigned long long* ent_used;
};
enum { S_HOST = 0, S_APP = 1, S_PROC = 2, S_MSGID = 3, S_MSG = 4, S_FULL = 5 };

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
};
struct GlobalReader {
    const uint32_t* words;  // packed buffer viewed as dwords (base is 16-byte aligned)
    uint64_t base;          // byte offset of the line inside the packed buffer
    uint64_t cur_idx = ~0ull;
    uint32_t cur = 0;
    __device__ __forceinline__ GlobalReader(const uint32_t* w, uint64_t b) : words(w), base(b) {}
    __device__ __forceinline__ uint32_t byte(uint32_t i) {
        uint64_t a = base + i;
        uint64_t w = a >> 2;
        if (w != cur_idx) {
            cur = words[w];
            cur_idx = w;
        }
        return __builtin_amdgcn_ubfe(cur, ((uint32_t)a & 3u) * 8u, 8u);
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

// ---------------------------------------------------------------------------------------------
// Calendar + timestamp arithmetic (time 0.3: Date::from_calendar_date, Time::from_hms_nano,
// UtcOffset::from_hms, OffsetDateTime::unix_timestamp_nanos).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool is_leap_year(int y) { return (y % 4 == 0) && (y % 100 != 0 || y % 400 == 0); }
__device__ __forceinline__ int days_in_month(int y, int m) {
    // 31 28 31 30 31 30 31 31 30 31 30 31 packed 2 bits each above 28
    int d = 28 + ((0xEEFBB3 >> ((m - 1) * 2)) & 3);  // Jan..Dec extra days: 3 0 3 2 3 2 3 3 2 3 2 3
    return (m == 2 && is_leap_year(y)) ? 29 : d;
}
__device__ __forceinline__ int64_t days_from_civil(int y, int m, int d) {
    y -= m <= 2;
    int era = (y >= 0 ? y : y - 399) / 400;
    int yoe = y - era * 400;
    int doy = (153 * (m + (m > 2 ? -3 : 9)) + 2) / 5 + d - 1;
    int doe = yoe * 365 + yoe / 4 - yoe / 100 + doy;
    return (int64_t)era * 146097 + doe - 719468;
}
__device__ __forceinline__ void civil_from_days(int64_t z, int* y, int* m, int* d) {
    z += 719468;
    int64_t era = (z >= 0 ? z : z - 146096) / 146097;
    int doe = (int)(z - era * 146097);
    int yoe = (doe - doe / 1460 + doe / 36524 - doe / 146096) / 365;
    int yy = yoe + (int)era * 400;
    int doy = doe - (365 * yoe + yoe / 4 - yoe / 100);
    int mp = (5 * doy + 2) / 153;
    *d = doy - (153 * mp + 2) / 5 + 1;
    *m = mp < 10 ? mp + 3 : mp - 9;
    *y = yy + (*m <= 2);
}
// (secs * 1e9 + nano) as i128 -> f64 (RNE) -> / 1e9, without __int128 runtime support.
__device__ __forceinline__ double unix_nanos_to_f64(int64_t secs, uint32_t nano) {
    // total = secs*1e9 + nano ; nano in [0, 1e9).  Work on the magnitude.
    bool neg = secs < 0;
    uint64_t lo, hi;
    if (!neg) {
        uint64_t a = (uint64_t)secs;
        lo = a * 1000000000ull;
        hi = __umul64hi(a, 1000000000ull);
        uint64_t l2 = lo + nano;
        hi += l2 < lo;
        lo = l2;
    } else {
        // |total| = (-secs)*1e9 - nano   (secs <= -1 so this is > 0)
        uint64_t a = (uint64_t)(-secs);
        lo = a * 1000000000ull;
        hi = __umul64hi(a, 1000000000ull);
        uint64_t l2 = lo - nano;
        hi -= l2 > lo;
        lo = l2;
    }
    double mag;
    if (hi == 0) {
        mag = (double)lo;  // u64 -> f64 is correctly rounded (RNE)
    } else {
        // keep 64 significant bits, fold the shifted-out bits into a sticky LSB: rounding a
        // 64-bit integer to 53 bits then sees exactly the same round/sticky information.
        int s = 64 - __clzll((long long)hi);  // 1..64; here hi < 2^5
        uint64_t m = (hi << (64 - s)) | (lo >> s);
        uint64_t lost = lo & ((1ull << s) - 1ull);
        m |= (lost != 0);
        mag = ldexp((double)m, s);  // exact scaling
    }
    double f = neg ? -mag : mag;
    return f / 1e9;  // IEEE-754 correctly rounded division (no fast-math)
}

struct DateTimeParts {
    int year, month, day, hour, minute, second;
    uint32_t nano;
    int off_sign, off_h, off_m;
};
// Validation + conversion; allow_leap = the Rfc3339 parser's second==60 stand-in.
__device__ __forceinline__ bool datetime_to_unix(const DateTimeParts& p, bool allow_leap, double* out) {
    int second = p.second;
    uint32_t nano = p.nano;
    bool leap = false;
    if (second == 60 && allow_leap) {
        second = 59;
        nano = 999999999u;
        leap = true;
    }
    if (p.month < 1 || p.month > 12) return false;
    if (p.year < -9999 || p.year > 9999) return false;
    if (p.day < 1 || p.day > days_in_month(p.year, p.month)) return false;
    if (p.hour > 23 || p.minute > 59 || second > 59) return false;
    if (p.off_h > 25 || p.off_m > 59) return false;
    int off = p.off_sign * (p.off_h * 3600 + p.off_m * 60);
    int64_t secs = days_from_civil(p.year, p.month, p.day) * 86400 + (p.hour * 3600 + p.minute * 60 + second - off);
    if (leap) {
        int64_t days = secs >= 0 ? secs / 86400 : -((-secs + 86399) / 86400);
        int64_t sod = secs - days * 86400;
        int y, m, d;
        civil_from_days(days, &y, &m, &d);
        if (sod != 86399 || d != days_in_month(y, m)) return false;
    }
    *out = unix_nanos_to_f64(secs, nano);
    return true;
}

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
__device__ __forceinline__ void stage_tile(const uint8_t* __restrict__ bytes, uint64_t a0, uint32_t span, uint8_t* smem) {
    const uint4* __restrict__ src = reinterpret_cast<const uint4*>(bytes + a0);
    uint4* dst = reinterpret_cast<uint4*>(smem);
    const uint32_t nchunk = span >> 4;
    const uint32_t lane = threadIdx.x;
    for (uint32_t c0 = 0; c0 < nchunk; c0 += kWave * 8) {
        uint4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            uint32_t idx = c0 + k * kWave + lane;
            if (idx < nchunk) v[k] = src[idx];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            uint32_t idx = c0 + k * kWave + lane;
            if (idx < nchunk) dst[idx] = v[k];
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
