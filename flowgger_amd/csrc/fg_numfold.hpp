// fg_numfold.hpp -- decimal tokens of the everyday shape  -?D+(.D+)?  (<= 19 digits, <= 24 bytes) folded to a 64-bit significand
// straight out of registers, for the decoders that convert numbers on the GPU (GELF: serde_json 0.8 numbers; LTSV: Rust
// `u64 / i64 / f64::from_str` and the float form of the timestamp).  No loop over digits, no memory round trip per digit:
// class masks by SWAR, every dword's digits folded with ONE v_dot4_u32_u8 against a weight vector looked up by the dword's 4-bit
// digit mask (a '.' or the sign inside the dword gets weight 0), the six partial values chained with 24-bit pair multiplies
// and two wide ones.  A token that is not of that shape is NOT judged here: the caller runs the exact byte-wise parser of its
// grammar (fg_numparse.hpp), which owns every error and every other spelling (exponents, 20+ digits, "+5", ".5", "inf").
//
// Host + device (written against fg_wave.hpp): the CPU suite checks the fold against the oracle's number scanners
// (tests/test_wave_gelf_cpu.py::test_register_number_parser_matches_serde_json).
#pragma once
#include <stdint.h>

#include "fg_wave.hpp"

namespace fg {
namespace numfold {

constexpr uint32_t kDwWords = 28;   // [0..15] digit weights by 4-bit digit mask, [16..24] 10^0 .. 10^8, rest unused
constexpr uint32_t kP10Words = 23;  // 10^0 .. 10^22 as doubles (exact)
constexpr uint32_t kTableBytes = kP10Words * 8u + kDwWords * 4u;  // p10 first (8-byte aligned), then dw

// once per wave, by all lanes; the caller synchronises afterwards
FG_WV void init_tables(uint32_t* dw, double* p10) {
    const uint32_t lane = wv::lane();
    if (lane == 0) {
        double v = 1.0;
        for (uint32_t k = 0; k < kP10Words; ++k) {
            p10[k] = v;
            v *= 10.0;
        }
    }
    if (lane < 16u) {
        // byte i of entry m: 10^(digit bytes above i) when bit i of m is set, else 0 -- except 1000 (m = 15, byte 0), which does
        // not fit a byte and is added separately
        uint32_t wgt = 0, above = 0;
        for (int i = 3; i >= 0; --i) {
            if ((lane >> i) & 1u) {
                const uint32_t p = above == 0 ? 1u : above == 1 ? 10u : above == 2 ? 100u : 0u;
                wgt |= p << (8 * i);
                ++above;
            }
        }
        dw[lane] = wgt;
    }
    if (lane >= 16u && lane < 25u) {
        uint32_t p = 1;
        for (uint32_t k = 16u; k < lane; ++k) p *= 10u;
        dw[lane] = p;
    }
}

struct Folded {
    bool ok;        // the token is  -?D+(.D+)?  with 1..19 digits (leading zeros count and are allowed: the caller's grammar decides)
    bool neg, has_dot;
    uint32_t ni, nf;  // digits before / behind the '.'
    uint32_t c0;      // the first digit's byte
    uint64_t sig;     // all digits as one integer (< 10^19 < 2^64)
};

// w = the token's first 24 bytes (little endian, bytes beyond n are ignored), n <= 24, dwt = the dw table
FG_WV Folded fold24(const uint32_t w[6], uint32_t n, const uint32_t* dwt) {
    Folded f;
    f.neg = (w[0] & 0xFFu) == '-';
    uint32_t x[6], ndm = 0;
#pragma unroll
    for (uint32_t k = 0; k < 6; ++k) {
        x[k] = w[k] ^ 0x30303030u;
        const uint32_t nd = (x[k] | ((x[k] & 0x7F7F7F7Fu) + 0x76767676u)) & 0x80808080u;  // bit 7: the byte is not a digit
        ndm |= wv::udot4(nd >> 7, 0x08040201u, 0u) << (4u * k);
    }
    const uint32_t tok = n >= 24u ? 0xFFFFFFu : (1u << n) - 1u;
    const uint32_t o = f.neg ? 1u : 0u;
    const uint32_t body = tok & ~o;
    const uint32_t nondig = ndm & body, digits = body & ~ndm;
    const uint32_t nd_total = wv::popc32(digits);
    f.has_dot = nondig != 0u;
    const uint32_t dp = f.has_dot ? wv::ctz32(nondig) : n;
    f.ni = dp - o;
    f.nf = f.has_dot ? n - dp - 1u : 0u;
    f.c0 = f.neg ? ((w[0] >> 8) & 0xFFu) : (w[0] & 0xFFu);
    // the one non-digit must be a '.': its byte, picked out of the six dwords
    uint32_t dsel = w[0];
#pragma unroll
    for (uint32_t k = 1; k < 6; ++k) dsel = (dp >> 2) == k ? w[k] : dsel;
    const bool dot_ok = !f.has_dot || ((dsel >> (8u * (dp & 3u))) & 0xFFu) == '.';
    f.ok = n >= 1u && n <= 24u && (nondig & (nondig - 1u)) == 0u && dot_ok && nd_total >= 1u && nd_total <= 19u && dp >= o + 1u &&
           (!f.has_dot || f.nf >= 1u);
    // the dwords' values (<= 9999 each) pair up with 24-bit multiplies (full rate); two wide multiplies chain the three pairs
    uint32_t val[6], cnt[6];
#pragma unroll
    for (uint32_t k = 0; k < 6; ++k) {
        const uint32_t m = (digits >> (4u * k)) & 15u;
        val[k] = wv::udot4(x[k], dwt[m], m == 15u ? (x[k] & 0xFFu) * 1000u : 0u);
        cnt[k] = wv::popc32(m);
    }
    const uint32_t* p10u = dwt + 16;
    const uint32_t p0 = wv::mad24(val[0], p10u[cnt[1]], val[1]);  // < 10^8
    const uint32_t p1 = wv::mad24(val[2], p10u[cnt[3]], val[3]);
    const uint32_t p2 = wv::mad24(val[4], p10u[cnt[5]], val[5]);
    uint64_t sig = (uint64_t)p0 * p10u[cnt[2] + cnt[3]] + p1;  // < 10^16
    sig = sig * p10u[cnt[4] + cnt[5]] + p2;                   // <= 19 digits: < 2^64
    f.sig = sig;
    return f;
}

}  // namespace numfold
}  // namespace fg
