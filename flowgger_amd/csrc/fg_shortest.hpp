// fg_shortest.hpp -- f64 -> text exactly as Rust's `impl Display for f64` (`{}` / `to_string()`):
// the SHORTEST decimal digit string that reads back as the same double (closest to the true value
// when several have that length), laid out without an exponent (core::fmt::float:
// float_to_decimal_display -> flt2dec::to_shortest_str with frac_digits = 0 -> digits_to_dec_str).
//
// Where the reference prints an f64 this way: LTSVEncoder `time:{ts}` and F64 pair values
// (encoder/ltsv_encoder.rs:86,106), `impl Display for StructuredData` used by the RFC5424 / RFC3164
// encoders (record.rs:58).  (The GELF encoder goes through serde_json -> the dtoa crate = Grisu2 with
// a different layout: fg_dtoa.hpp.)
//
// Digits: the Schubfach algorithm (R. Giulietti) -- one 128-bit power of ten per value, three
// 64x128-bit multiplications, no fallback path; produces the same digits as Rust's Grisu3 + Dragon4
// (both are "shortest, then closest").  Host + device; checked on the CPU against libstdc++'s
// std::to_chars (Ryu) over 10^8-scale random and structured inputs (tests/test_shortest_cpu.py).
#pragma once
#include "fg_dtoa.hpp"
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define FGS_HD __host__ __device__ __forceinline__
#else
#define FGS_HD inline
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define FGS_TABLE static __device__ const
#else
#define FGS_TABLE static const
#endif

namespace fg {
namespace shortest {

#include "fg_shortest_table.inc"

struct Dec {
    uint64_t sig;  // decimal significand without trailing zeros (0 only for the value zero)
    int32_t exp;   // value = sig * 10^exp
};

FGS_HD void mul64(uint64_t a, uint64_t b, uint64_t* hi, uint64_t* lo) {
#if defined(__HIP_DEVICE_COMPILE__)
    *lo = a * b;
    *hi = __umul64hi(a, b);
#else
    const unsigned __int128 p = (unsigned __int128)a * b;
    *lo = (uint64_t)p;
    *hi = (uint64_t)(p >> 64);
#endif
}
// floor((g * cp) / 2^128) with the sticky bit of the discarded part or-ed into bit 0
FGS_HD uint64_t round_to_odd(uint64_t ghi, uint64_t glo, uint64_t cp) {
    uint64_t xh, xl, yh, yl;
    mul64(glo, cp, &xh, &xl);
    mul64(ghi, cp, &yh, &yl);
    const uint64_t y0 = yl + xh;
    const uint64_t y1 = yh + (y0 < yl ? 1u : 0u);
    return y1 | (y0 > 1u ? 1u : 0u);
}
FGS_HD int32_t floor_log2_pow10(int32_t e) { return (int32_t)(((int64_t)e * 1741647) >> 19); }

// bits = raw IEEE-754 bits of a finite, non-zero double (sign ignored)
FGS_HD Dec to_decimal(uint64_t bits) {
    const uint64_t frac = bits & 0x000FFFFFFFFFFFFFull;
    const uint32_t bexp = (uint32_t)(bits >> 52) & 0x7FFu;
    uint64_t c;
    int32_t q;
    uint64_t sig;
    int32_t k;
    bool done = false;
    if (bexp != 0) {
        c = frac | 0x0010000000000000ull;
        q = (int32_t)bexp - 1075;
        if (q <= 0 && q > -53 && (c & ((1ull << -q) - 1u)) == 0) {  // an integer below 2^53
            sig = c >> -q;
            k = 0;
            done = true;
        }
    } else {
        c = frac;
        q = -1074;
    }
    if (!done) {
        const bool even = (c & 1u) == 0;
        const bool lower_closer = frac == 0 && bexp > 1;
        const uint64_t cbl = 4 * c - 2 + (lower_closer ? 1u : 0u);
        const uint64_t cb = 4 * c;
        const uint64_t cbr = 4 * c + 2;
        // floor(log10(2^q)) resp. floor(log10(3/4 * 2^q))
        k = (int32_t)(((int64_t)q * 1262611 - (lower_closer ? 524031 : 0)) >> 22);
        const int32_t h = q + floor_log2_pow10(-k) + 1;  // 1..4
        const uint64_t ghi = kPow10_128[-k + 292][0], glo = kPow10_128[-k + 292][1];
        const uint64_t vbl = round_to_odd(ghi, glo, cbl << h);
        const uint64_t vb = round_to_odd(ghi, glo, cb << h);
        const uint64_t vbr = round_to_odd(ghi, glo, cbr << h);
        const uint64_t lower = vbl + (even ? 0u : 1u);
        const uint64_t upper = vbr - (even ? 0u : 1u);
        const uint64_t s = vb >> 2;
        bool found = false;
        if (s >= 10) {
            const uint64_t sp = s / 10;
            const bool up_inside = lower <= 40 * sp;
            const bool wp_inside = 40 * sp + 40 <= upper;
            if (up_inside != wp_inside) {
                sig = sp + (wp_inside ? 1u : 0u);
                k += 1;
                found = true;
            }
        }
        if (!found) {
            const bool u_inside = lower <= 4 * s;
            const bool w_inside = 4 * s + 4 <= upper;
            if (u_inside != w_inside) {
                sig = s + (w_inside ? 1u : 0u);
            } else {
                const uint64_t mid = 4 * s + 2;
                const bool round_up = vb > mid || (vb == mid && (s & 1u) != 0);
                sig = s + (round_up ? 1u : 0u);
            }
        }
    }
    while (sig != 0 && sig % 10u == 0) {
        sig /= 10u;
        ++k;
    }
    return Dec{sig, k};
}

// `format!("{}", v)`: put(c) receives the bytes ("NaN", "inf", "-inf", "-0", "1438790025.637824",
// "0.000001", "100000000000000000000000"); returns nothing, the sink counts.
template <class S>
FGS_HD void display_f64(double v, S& out) {
    uint64_t bits;
    memcpy(&bits, &v, 8);
    const bool neg = (bits >> 63) != 0;
    const uint64_t mag = bits & 0x7FFFFFFFFFFFFFFFull;
    if (mag > 0x7FF0000000000000ull) {  // NaN: no sign
        out.put('N');
        out.put('a');
        out.put('N');
        return;
    }
    if (neg) out.put('-');
    if (mag == 0x7FF0000000000000ull) {
        out.put('i');
        out.put('n');
        out.put('f');
        return;
    }
    if (mag == 0) {
        out.put('0');
        return;
    }
    const Dec d = to_decimal(mag);
    // the digits stay in registers (17 BCD nibbles): a char buffer would be scratch memory on the GPU
    int nd = 1;
    for (uint64_t p = 10u; d.sig >= p && nd < 19; p *= 10u) ++nd;
    dtoa::Digits dg;
    dg.v = d.sig;
    dg.len = nd;
    const dtoa::Bcd17 bcd(dg);
    // value = 0.d1d2... * 10^e10
    const int e10 = d.exp + nd;
    if (e10 > 0 && e10 <= 16 && nd <= 17) {
        // the point inside the first 17 positions (seconds with a fraction), or an integer of up to 16 digits (whole seconds, typed
        // values): assembled in registers, one piece or two (dtoa::put_text17) instead of a put per character
        const uint32_t L = nd > e10 ? (uint32_t)nd + 1u : (uint32_t)e10;
        if (S::kCount) out.add(L);
        else dtoa::put_text17(bcd, nd > e10 ? (uint32_t)e10 : 32u, L, out);
        return;
    }
    if (e10 <= 0) {
        out.put('0');
        out.put('.');
        for (int i = 0; i < -e10; ++i) out.put('0');
    }
    for (int i = 0; i < nd; ++i) {
        if (e10 > 0 && i == e10) out.put('.');
        out.put((uint32_t)'0' + bcd.digit(i));
    }
    for (int i = nd; i < e10; ++i) out.put('0');
}

}  // namespace shortest
}  // namespace fg
