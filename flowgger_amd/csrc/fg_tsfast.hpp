// fg_tsfast.hpp -- timestamps of the everyday spellings parsed straight out of registers (device only).
//
//   fast_rfc3339_core   "YYYY-MM-DD[Tt]HH:MM:SS[.d+](Z|z|[+-]HH:MM)"   time::OffsetDateTime::parse(.., &Rfc3339)
//                       (rfc5424_decoder.rs:94-99, ltsv_decoder.rs:224-229)
//   fast_english        "D{1,2}/Mon/YYYY:HH:MM:SS[.d{1,9}] [+-]HHMM"    the two "English" descriptions of ltsv_decoder.rs:236-254
//
// Both take the timestamp's first 36 bytes as nine dwords (little endian, byte 0 = first byte of the timestamp) and its length:
// SWAR digit / literal validation (XOR with the expected pattern, every byte checked against its limit), no loop, no memory
// round trip per byte -- the byte-wise parsers (fg_device.hpp parse_rfc3339, fg_ltsv.hip english_one) cost a dependent LDS read
// per byte.  They ACCEPT only what the byte-wise parsers accept with the same result (same field arithmetic:
// datetime_to_unix_fast, fg_timeconv.hpp); everything they do not decide goes to the byte-wise parser, which owns every error.
#pragma once
#include <stdint.h>

#include "fg_timeconv.hpp"

namespace fg {

// every byte of x (already XORed with the expected pattern) must be <= its limit, where
// add = 0x7F - limit per byte: returns nonzero iff some byte exceeds its limit.
__device__ __forceinline__ uint32_t swar_exceeds(uint32_t x, uint32_t add) {
    return (x | ((x & 0x7F7F7F7Fu) + add)) & 0x80808080u;
}
// four ASCII-digit values (0..9 per byte, first digit in the low byte) -> 0..9999
__device__ __forceinline__ uint32_t digits4(uint32_t x) {
    uint32_t t = (x * 10u + (x >> 8)) & 0x00FF00FFu;  // byte0 = 10*d0+d1, byte2 = 10*d2+d3
    return (t & 0xFFu) * 100u + (t >> 16);
}
// leading ASCII digits of four dwords already XORed with 0x30303030 (0..16)
__device__ __forceinline__ uint32_t leading_digits16(uint32_t f0, uint32_t f1, uint32_t f2, uint32_t f3) {
    const uint32_t n0 = swar_exceeds(f0, 0x76767676u), n1 = swar_exceeds(f1, 0x76767676u);
    const uint32_t n2 = swar_exceeds(f2, 0x76767676u), n3 = swar_exceeds(f3, 0x76767676u);
    return n0 ? (uint32_t)__builtin_ctz(n0) >> 3
              : n1 ? 4u + ((uint32_t)__builtin_ctz(n1) >> 3)
                   : n2 ? 8u + ((uint32_t)__builtin_ctz(n2) >> 3) : n3 ? 12u + ((uint32_t)__builtin_ctz(n3) >> 3) : 16u;
}
// the first min(nd, 9) fraction digits as nanoseconds (f0, f1, f2 = digit values, first digit in the low byte of f0)
__device__ __forceinline__ uint32_t nanos9(uint32_t f0, uint32_t f1, uint32_t f2, uint32_t nd) {
    const uint32_t keep = nd < 9u ? nd : 9u;
    const uint32_t k0 = keep >= 4u ? 0xFFFFFFFFu : (1u << (8u * keep)) - 1u;
    const uint32_t k1 = keep >= 8u ? 0xFFFFFFFFu : keep <= 4u ? 0u : (1u << (8u * (keep - 4u))) - 1u;
    const uint32_t d8 = keep >= 9u ? (f2 & 0xFFu) : 0u;
    return (digits4(f0 & k0) * 10000u + digits4(f1 & k1)) * 10u + d8;
}

// RFC3339, branch-free.  r = the timestamp's bytes 0..35, L = its length, ld8(pos, &lo, &hi) = eight bytes at its byte pos
// (the zone designator sits at a data-dependent position).  Returns 1 = converted, 0 = invalid, 2 = undecided (the caller runs
// the byte-wise parser).
template <class Load8>
__device__ __forceinline__ uint32_t fast_rfc3339_core(const uint32_t r[9], uint32_t L, Load8 ld8, double* out) {
    // bytes 0..18 = "YYYY-MM-DDtHH:MM:SS" ; XOR with the pattern: digits -> 0..9, literals -> 0
    const uint32_t x0 = r[0] ^ 0x30303030u;                         // Y Y Y Y
    const uint32_t x1 = r[1] ^ 0x2D30302Du;                         // - M M -
    const uint32_t x2 = (r[2] | 0x00200000u) ^ 0x30743030u;         // D D t H   ('T'|0x20 == 't')
    const uint32_t x3 = r[3] ^ 0x30303A30u;                         // H : M M
    const uint32_t x4 = (r[4] & 0x00FFFFFFu) ^ 0x0030303Au;         // : S S (byte 19 cleared)
    uint32_t bad = swar_exceeds(x0, 0x76767676u) | swar_exceeds(x1, 0x7F76767Fu) | swar_exceeds(x2, 0x767F7676u) |
                   swar_exceeds(x3, 0x76767F76u) | swar_exceeds(x4, 0x7F76767Fu);
    bad |= L < 20u ? 1u : 0u;
    DateTimeParts p;
    p.year = (int)digits4(x0);
    p.month = (int)(((x1 >> 8) & 0xFFu) * 10u + ((x1 >> 16) & 0xFFu));
    p.day = (int)((x2 & 0xFFu) * 10u + ((x2 >> 8) & 0xFFu));
    p.hour = (int)((x2 >> 24) * 10u + (x3 & 0xFFu));
    p.minute = (int)(((x3 >> 16) & 0xFFu) * 10u + (x3 >> 24));
    p.second = (int)(((x4 >> 8) & 0xFFu) * 10u + ((x4 >> 16) & 0xFFu));
    // optional fraction: digits live in bytes 20..35 = r[5..8]; count the leading digits
    const bool has_frac = (r[4] >> 24) == '.';
    const uint32_t f0 = r[5] ^ 0x30303030u, f1 = r[6] ^ 0x30303030u, f2 = r[7] ^ 0x30303030u, f3 = r[8] ^ 0x30303030u;
    uint32_t nd = leading_digits16(f0, f1, f2, f3);
    const uint32_t room = L - 20u;  // (garbage when L < 20: `bad` is already set)
    nd = nd > room ? room : nd;
    bad |= (has_frac && nd == 0u) ? 1u : 0u;
    const bool undecided = has_frac && nd >= 16u;  // very long fraction: the byte-wise parser decides
    // keep the first min(nd,9) digits, zero the rest => nine digits with trailing zeros
    p.nano = has_frac ? nanos9(f0, f1, f2, nd) : 0u;
    const uint32_t pos = has_frac ? 20u + nd : 19u;  // index of the time-zone designator
    uint32_t bad_tz = pos >= L ? 1u : 0u;            // (judged only when the fraction was decided here)
    // time-zone designator: re-read 8 bytes at its (data-dependent) position
    uint32_t z0, z1;
    ld8(pos, &z0, &z1);
    const uint32_t c = z0 & 0xFFu;
    const bool is_z = (c | 0x20u) == 'z';
    const bool is_off = c == '+' || c == '-';
    // bytes 1..5 = H H : M M
    const uint32_t y0 = (z0 >> 8) ^ 0x003A3030u;   // H H :   (3 bytes)
    const uint32_t y1 = (z1 & 0xFFFFu) ^ 0x3030u;  // M M
    const uint32_t off_bad = swar_exceeds(y0, 0x7F7F7676u) | swar_exceeds(y1, 0x7F7F7676u);
    bad_tz |= is_z ? (pos + 1u != L ? 1u : 0u) : is_off ? ((pos + 6u != L ? 1u : 0u) | off_bad) : 1u;
    p.off_sign = c == '-' ? -1 : 1;
    p.off_h = is_off ? (int)((y0 & 0xFFu) * 10u + ((y0 >> 8) & 0xFFu)) : 0;
    p.off_m = is_off ? (int)((y1 & 0xFFu) * 10u + ((y1 >> 8) & 0xFFu)) : 0;
    const uint32_t conv = (uint32_t)datetime_to_unix_fast(p, out);
    return bad ? 0u : undecided ? 2u : bad_tz ? 0u : conv;
}

// "English": D{1,2}/Mon/YYYY:HH:MM:SS[.d{1,9}] [+-]HHMM, the whole [0, L) consumed (both descriptions of ltsv_decoder.rs:236-254:
// the one without and the one with the subsecond).  r = the timestamp's bytes 0..35 (bytes beyond L: anything).
// true = converted exactly as english_one does; false = NOT DECIDED (a year sign, ten or more fraction digits, a leap second, a
// date outside 1970..2514, anything malformed): the caller runs the byte-wise chain.
__device__ __forceinline__ bool fast_english(const uint32_t r[9], uint32_t L, double* out) {
    const uint32_t b1 = (r[0] >> 8) & 0xFFu, b2 = (r[0] >> 16) & 0xFFu;
    const uint32_t d0 = (r[0] & 0xFFu) - '0', d1 = b1 - '0';
    const bool one = b1 == '/';
    const bool two = d1 <= 9u && b2 == '/';
    bool ok = d0 <= 9u && (one || two);
    const uint32_t s = one ? 2u : 3u;  // bytes in front of the month name
    // q[i] = bytes s + 4i .. : "Mon/" "YYYY" ":HH:" "MM:S" "S???" ...
    uint32_t q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) q[i] = one ? __builtin_amdgcn_alignbyte(r[i + 1], r[i], 2) : __builtin_amdgcn_alignbyte(r[i + 1], r[i], 3);
    DateTimeParts p;
    p.day = (int)(one ? d0 : d0 * 10u + d1);
    // month names, case-sensitive: Jan Feb Mar Apr May Jun Jul Aug Sep Oct Nov Dec, then '/'
    const uint32_t names[12] = {0x2F6E614Au, 0x2F626546u, 0x2F72614Du, 0x2F727041u, 0x2F79614Du, 0x2F6E754Au,
                                0x2F6C754Au, 0x2F677541u, 0x2F706553u, 0x2F74634Fu, 0x2F766F4Eu, 0x2F636544u};
    p.month = 0;
#pragma unroll
    for (int k = 0; k < 12; ++k) p.month = q[0] == names[k] ? k + 1 : p.month;
    ok = ok && p.month != 0;
    const uint32_t x1 = q[1] ^ 0x30303030u;                  // Y Y Y Y   (a sign: not decided here)
    const uint32_t x2 = q[2] ^ 0x3A30303Au;                  // : H H :
    const uint32_t x3 = q[3] ^ 0x303A3030u;                  // M M : S
    const uint32_t x4 = (q[4] & 0xFFu) ^ 0x30u;              // S
    const uint32_t bad = swar_exceeds(x1, 0x76767676u) | swar_exceeds(x2, 0x7F76767Fu) | swar_exceeds(x3, 0x767F7676u) |
                         swar_exceeds(x4, 0x7F7F7F76u);
    ok = ok && bad == 0u;
    p.year = (int)digits4(x1);
    p.hour = (int)(((x2 >> 8) & 0xFFu) * 10u + ((x2 >> 16) & 0xFFu));
    p.minute = (int)((x3 & 0xFFu) * 10u + ((x3 >> 8) & 0xFFu));
    p.second = (int)((x3 >> 24) * 10u + (x4 & 0xFFu));
    // byte s + 17: ' ' (no subsecond) or '.' + 1..9 digits + ' '
    const uint32_t c17 = (q[4] >> 8) & 0xFFu;
    const bool has_frac = c17 == '.';
    // fraction digits start at q-byte 18: f = bytes 18..33 of the q stream
    const uint32_t f0 = __builtin_amdgcn_alignbyte(q[5], q[4], 2) ^ 0x30303030u, f1 = __builtin_amdgcn_alignbyte(q[6], q[5], 2) ^ 0x30303030u;
    const uint32_t f2 = __builtin_amdgcn_alignbyte(q[7], q[6], 2) ^ 0x30303030u;
    const uint32_t nd = has_frac ? leading_digits16(f0, f1, f2, 0xFFFFFFFFu) : 0u;
    ok = ok && (has_frac ? (nd >= 1u && nd <= 9u) : c17 == ' ');
    p.nano = has_frac ? nanos9(f0, f1, f2, nd) : 0u;
    // " +HHMM" = six bytes at q-byte sp, the last of the timestamp
    const uint32_t sp = has_frac ? 18u + nd : 17u;  // <= 27
    ok = ok && s + sp + 6u == L;
    // the six bytes out of q[4..7] (sp in 17..27 -> dword 4..6)
    const uint32_t di = sp >> 2, sh = sp & 3u;
    const uint32_t a0 = di == 4u ? q[4] : di == 5u ? q[5] : q[6];
    const uint32_t a1 = di == 4u ? q[5] : di == 5u ? q[6] : q[7];
    const uint32_t a2 = di == 4u ? q[6] : di == 5u ? q[7] : 0u;
    uint32_t z0, z1;
    if (sh == 0u) { z0 = a0; z1 = a1; }
    else if (sh == 1u) { z0 = __builtin_amdgcn_alignbyte(a1, a0, 1); z1 = __builtin_amdgcn_alignbyte(a2, a1, 1); }
    else if (sh == 2u) { z0 = __builtin_amdgcn_alignbyte(a1, a0, 2); z1 = __builtin_amdgcn_alignbyte(a2, a1, 2); }
    else { z0 = __builtin_amdgcn_alignbyte(a1, a0, 3); z1 = __builtin_amdgcn_alignbyte(a2, a1, 3); }
    const uint32_t sgn = (z0 >> 8) & 0xFFu;
    ok = ok && (z0 & 0xFFu) == ' ' && (sgn == '+' || sgn == '-');
    const uint32_t y0 = ((z0 >> 16) | (z1 << 16)) ^ 0x30303030u;  // H H M M
    ok = ok && swar_exceeds(y0, 0x76767676u) == 0u;
    p.off_sign = sgn == '-' ? -1 : 1;
    p.off_h = (int)((y0 & 0xFFu) * 10u + ((y0 >> 8) & 0xFFu));
    p.off_m = (int)(((y0 >> 16) & 0xFFu) * 10u + (y0 >> 24));
    double v = 0.0;
    const int conv = datetime_to_unix_fast(p, &v);  // 1 = converted; 0 (invalid field) / 2 (outside its domain): not decided here
    *out = v;
    return ok && conv == 1;
}

}  // namespace fg
