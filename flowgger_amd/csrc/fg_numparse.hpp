// fg_numparse.hpp -- decimal -> binary number parsing with the EXACT semantics of the reference's
// dependencies, usable from host (unit-fuzzed against strtod on the CPU) and device code.
//
//   * Rust `str::parse::<f64>` = core::num::dec2flt (ltsv_decoder.rs:157,256-261): correctly
//     rounded.  Restated here in its three published stages:
//       1. Clinger fast path (<= 19 digits, |exp10| small, exact f64 multiply / divide),
//       2. Eisel-Lemire with the 128-bit power-of-five table,
//       3. the "Decimal" big-digit slow path (768 digits + truncated flag), used when (2)
//          cannot decide or when more than 19 significant digits straddle a rounding boundary.
//   * Rust integer / bool `from_str` (ltsv_decoder.rs:116,142,174,190).
//   * serde_json 0.8 number scanning: u64 significand, then `f *=` / `f /= POW10[|e|]`
//     (deliberately NOT correctly rounded; gelf_decoder.rs:42,53,93-95).
//
// A reader `R` only needs `uint32_t byte(uint32_t i)`.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define FG_HD __host__ __device__ __forceinline__
#define FG_HDN __host__ __device__ inline
#else
#define FG_HD inline
#define FG_HDN inline
#endif

namespace fg {
namespace num {

#if defined(__HIP_DEVICE_COMPILE__)
#define FG_TABLE_QUAL static __device__ const
#else
#define FG_TABLE_QUAL static const
#endif
#include "fg_numparse_tables.inc"
#undef FG_TABLE_QUAL

FG_HD double bits_to_f64(uint64_t b) {
    union { uint64_t u; double d; } x;
    x.u = b;
    return x.d;
}
FG_HD uint64_t f64_to_bits(double d) {
    union { uint64_t u; double d; } x;
    x.d = d;
    return x.u;
}
FG_HD void mul64(uint64_t a, uint64_t b, uint64_t* lo, uint64_t* hi) {
#if defined(__HIP_DEVICE_COMPILE__)
    *lo = a * b;
    *hi = __umul64hi(a, b);
#else
    unsigned __int128 p = (unsigned __int128)a * b;
    *lo = (uint64_t)p;
    *hi = (uint64_t)(p >> 64);
#endif
}
FG_HD int clz64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __clzll((long long)x);
#else
    return __builtin_clzll(x);
#endif
}
FG_HD bool is_digit(uint32_t c) { return (c - '0') <= 9u; }

// ---------------------------------------------------------------------------------------------
// integers and bool (Rust core::num::from_str_radix(10) / bool::from_str)
// ---------------------------------------------------------------------------------------------
// unsigned: optional '+', >= 1 digit, overflow above `max` -> false
template <class R>
FG_HD bool parse_unsigned(R& rd, uint32_t b, uint32_t e, uint64_t max, uint64_t* out) {
    if (b >= e) return false;
    if (rd.byte(b) == '+') ++b;
    if (b >= e) return false;
    uint64_t v = 0;
    for (; b < e; ++b) {
        uint32_t d = rd.byte(b) - '0';
        if (d > 9u) return false;
        if (v > (max - d) / 10u) return false;
        v = v * 10u + d;
    }
    *out = v;
    return true;
}
template <class R>
FG_HD bool parse_i64(R& rd, uint32_t b, uint32_t e, int64_t* out) {
    if (b >= e) return false;
    bool neg = false;
    uint32_t c = rd.byte(b);
    if (c == '+') ++b;
    else if (c == '-') { neg = true; ++b; }
    if (b >= e) return false;
    const uint64_t lim = neg ? (1ull << 63) : (1ull << 63) - 1ull;
    uint64_t v = 0;
    for (; b < e; ++b) {
        uint32_t d = rd.byte(b) - '0';
        if (d > 9u) return false;
        if (v > (lim - d) / 10u) return false;
        v = v * 10u + d;
    }
    *out = neg ? (int64_t)(0ull - v) : (int64_t)v;
    return true;
}
template <class R>
FG_HD bool bytes_equal(R& rd, uint32_t b, uint32_t e, const char* lit, uint32_t n) {
    if (e - b != n) return false;
    for (uint32_t i = 0; i < n; ++i)
        if (rd.byte(b + i) != (uint32_t)(uint8_t)lit[i]) return false;
    return true;
}
template <class R>
FG_HD bool bytes_equal_nocase(R& rd, uint32_t b, uint32_t e, const char* lower, uint32_t n) {
    if (e - b != n) return false;
    for (uint32_t i = 0; i < n; ++i)
        if ((rd.byte(b + i) | 0x20u) != (uint32_t)(uint8_t)lower[i]) return false;
    return true;
}

// ---------------------------------------------------------------------------------------------
// dec2flt stage 2: Eisel-Lemire.  Result is a biased (mantissa, power2); e < 0 means "undecided".
// ---------------------------------------------------------------------------------------------
struct BiasedFp {
    uint64_t f;
    int32_t e;
};
FG_HD bool same_fp(BiasedFp a, BiasedFp b) { return a.f == b.f && a.e == b.e; }

FG_HD void product_approx(int64_t q, uint64_t w, uint64_t* lo, uint64_t* hi) {
    const uint64_t mask = 0xFFFFFFFFFFFFFFFFull >> 55;  // precision = 52 + 3
    const int idx = (int)(q + 342);
    uint64_t first_lo, first_hi;
    mul64(w, FG_POW5_128[idx][0], &first_lo, &first_hi);
    if ((first_hi & mask) == mask) {
        uint64_t second_lo, second_hi;
        mul64(w, FG_POW5_128[idx][1], &second_lo, &second_hi);
        first_lo += second_hi;
        if (second_hi > first_lo) first_hi += 1;
    }
    *lo = first_lo;
    *hi = first_hi;
}
FG_HD BiasedFp compute_float(int64_t q, uint64_t w) {
    const BiasedFp zero{0, 0}, inf{0, 0x7FF}, err{0, -1};
    if (w == 0 || q < -342) return zero;
    if (q > 308) return inf;
    int lz = clz64(w);
    w <<= lz;
    uint64_t lo, hi;
    product_approx(q, w, &lo, &hi);
    if (lo == 0xFFFFFFFFFFFFFFFFull) {
        if (!(q >= -27 && q <= 55)) return err;
    }
    int upperbit = (int)(hi >> 63);
    uint64_t mantissa = hi >> (upperbit + 64 - 52 - 3);
    int32_t power2 = (int32_t)((((int32_t)q * (152170 + 65536)) >> 16) + 63) + upperbit - lz + 1023;
    if (power2 <= 0) {
        if (-power2 + 1 >= 64) return zero;
        mantissa >>= (-power2 + 1);
        mantissa += mantissa & 1u;
        mantissa >>= 1;
        power2 = mantissa >= (1ull << 52) ? 1 : 0;
        return BiasedFp{mantissa, power2};
    }
    if (lo <= 1 && q >= -4 && q <= 23 && (mantissa & 3u) == 1u && (mantissa << (upperbit + 64 - 52 - 3)) == hi) {
        mantissa &= ~1ull;
    }
    mantissa += mantissa & 1u;
    mantissa >>= 1;
    if (mantissa >= (2ull << 52)) {
        mantissa = 1ull << 52;
        power2 += 1;
    }
    mantissa &= ~(1ull << 52);
    if (power2 >= 0x7FF) return inf;
    return BiasedFp{mantissa, power2};
}

// ---------------------------------------------------------------------------------------------
// dec2flt stage 3: Decimal slow path (768 digits; caller supplies the digit buffer)
// ---------------------------------------------------------------------------------------------
constexpr uint32_t kDecMaxDigits = 768;
constexpr uint32_t kDecMaxDigitsNoOverflow = 19;
constexpr int32_t kDecPointRange = 2047;

struct Decimal {
    uint8_t* digits;  // kDecMaxDigits bytes
    uint32_t num_digits;
    int32_t decimal_point;
    bool truncated;
};
FG_HDN void dec_trim(Decimal& d) {
    while (d.num_digits != 0 && d.digits[d.num_digits - 1] == 0) d.num_digits -= 1;
}
FG_HDN uint64_t dec_round(const Decimal& d) {
    if (d.num_digits == 0 || d.decimal_point < 0) return 0;
    if (d.decimal_point > 18) return 0xFFFFFFFFFFFFFFFFull;
    uint32_t dp = (uint32_t)d.decimal_point;
    uint64_t n = 0;
    for (uint32_t i = 0; i < dp; ++i) {
        n *= 10;
        if (i < d.num_digits) n += d.digits[i];
    }
    bool round_up = false;
    if (dp < d.num_digits) {
        round_up = d.digits[dp] >= 5;
        if (d.digits[dp] == 5 && dp + 1 == d.num_digits) {
            round_up = d.truncated || (dp != 0 && (1 & d.digits[dp - 1]) != 0);
        }
    }
    return round_up ? n + 1 : n;
}
FG_HDN uint32_t dec_new_digits_lshift(const Decimal& d, uint32_t shift) {
    shift &= 63;
    uint32_t x_a = FG_LSHIFT_TABLE[shift], x_b = FG_LSHIFT_TABLE[shift + 1];
    uint32_t num_new = x_a >> 11;
    uint32_t pow5_a = x_a & 0x7FF, pow5_b = x_b & 0x7FF;
    for (uint32_t i = 0; i < pow5_b - pow5_a; ++i) {
        uint8_t p5 = FG_LSHIFT_POW5[pow5_a + i];
        if (i >= d.num_digits) return num_new - 1;
        if (d.digits[i] == p5) continue;
        return d.digits[i] < p5 ? num_new - 1 : num_new;
    }
    return num_new;
}
FG_HDN void dec_left_shift(Decimal& d, uint32_t shift) {
    if (d.num_digits == 0) return;
    uint32_t num_new = dec_new_digits_lshift(d, shift);
    uint32_t read = d.num_digits, write = d.num_digits + num_new;
    uint64_t n = 0;
    while (read != 0) {
        read -= 1;
        write -= 1;
        n += (uint64_t)d.digits[read] << shift;
        uint64_t quo = n / 10, rem = n - 10 * quo;
        if (write < kDecMaxDigits) d.digits[write] = (uint8_t)rem;
        else if (rem > 0) d.truncated = true;
        n = quo;
    }
    while (n > 0) {
        write -= 1;
        uint64_t quo = n / 10, rem = n - 10 * quo;
        if (write < kDecMaxDigits) d.digits[write] = (uint8_t)rem;
        else if (rem > 0) d.truncated = true;
        n = quo;
    }
    d.num_digits += num_new;
    if (d.num_digits > kDecMaxDigits) d.num_digits = kDecMaxDigits;
    d.decimal_point += (int32_t)num_new;
    dec_trim(d);
}
FG_HDN void dec_right_shift(Decimal& d, uint32_t shift) {
    uint32_t read = 0, write = 0;
    uint64_t n = 0;
    while ((n >> shift) == 0) {
        if (read < d.num_digits) {
            n = 10 * n + d.digits[read];
            read += 1;
        } else if (n == 0) {
            return;
        } else {
            while ((n >> shift) == 0) {
                n *= 10;
                read += 1;
            }
            break;
        }
    }
    d.decimal_point -= (int32_t)read - 1;
    if (d.decimal_point < -kDecPointRange) {
        d.num_digits = 0;
        d.decimal_point = 0;
        d.truncated = false;
        return;
    }
    const uint64_t mask = (1ull << shift) - 1;
    while (read < d.num_digits) {
        uint8_t new_digit = (uint8_t)(n >> shift);
        n = 10 * (n & mask) + d.digits[read];
        read += 1;
        d.digits[write] = new_digit;
        write += 1;
    }
    while (n > 0) {
        uint8_t new_digit = (uint8_t)(n >> shift);
        n = 10 * (n & mask);
        if (write < kDecMaxDigits) {
            d.digits[write] = new_digit;
            write += 1;
        } else if (new_digit > 0) {
            d.truncated = true;
        }
    }
    d.num_digits = write;
    dec_trim(d);
}
// parse_decimal: the (already grammar-checked, sign-stripped) text [b, e)
template <class R>
FG_HDN void dec_parse(R& rd, uint32_t b, uint32_t e, Decimal& d) {
    d.num_digits = 0;
    d.decimal_point = 0;
    d.truncated = false;
    uint32_t p = b;
    while (p < e && rd.byte(p) == '0') ++p;
    while (p < e && is_digit(rd.byte(p))) {
        if (d.num_digits < kDecMaxDigits) d.digits[d.num_digits] = (uint8_t)(rd.byte(p) - '0');
        d.num_digits += 1;
        ++p;
    }
    if (p < e && rd.byte(p) == '.') {
        ++p;
        uint32_t first = p;
        if (d.num_digits == 0) {
            while (p < e && rd.byte(p) == '0') ++p;  // leading zeros after the point
        }
        while (p < e && is_digit(rd.byte(p))) {
            if (d.num_digits < kDecMaxDigits) d.digits[d.num_digits] = (uint8_t)(rd.byte(p) - '0');
            d.num_digits += 1;
            ++p;
        }
        d.decimal_point = (int32_t)first - (int32_t)p;  // -(number of bytes after the point)
    }
    if (d.num_digits != 0) {
        // ignore trailing zeros of the digit string (walk back over '0' and '.')
        uint32_t n_trailing = 0;
        for (uint32_t q = p; q > b;) {
            --q;
            uint32_t c = rd.byte(q);
            if (c == '0') n_trailing += 1;
            else if (c != '.') break;
        }
        d.decimal_point += (int32_t)n_trailing;
        d.num_digits -= n_trailing;
        d.decimal_point += (int32_t)d.num_digits;
        if (d.num_digits > kDecMaxDigits) {
            d.truncated = true;
            d.num_digits = kDecMaxDigits;
        }
    }
    if (p < e && (rd.byte(p) | 0x20u) == 'e') {
        ++p;
        bool neg_exp = false;
        if (p < e && (rd.byte(p) == '-' || rd.byte(p) == '+')) {
            neg_exp = rd.byte(p) == '-';
            ++p;
        }
        int32_t exp_num = 0;
        while (p < e && is_digit(rd.byte(p))) {
            if (exp_num < 0x10000) exp_num = 10 * exp_num + (int32_t)(rd.byte(p) - '0');
            ++p;
        }
        d.decimal_point += neg_exp ? -exp_num : exp_num;
    }
    for (uint32_t i = d.num_digits; i < kDecMaxDigitsNoOverflow; ++i) d.digits[i] = 0;
}
template <class R>
FG_HDN BiasedFp parse_long_mantissa(R& rd, uint32_t b, uint32_t e, uint8_t* digit_buf) {
    const BiasedFp zero{0, 0}, inf{0, 0x7FF};
    const uint32_t kMaxShift = 60;
    const uint8_t powers[19] = {0, 3, 6, 9, 13, 16, 19, 23, 26, 29, 33, 36, 39, 43, 46, 49, 53, 56, 59};
    Decimal d;
    d.digits = digit_buf;
    dec_parse(rd, b, e, d);
    if (d.num_digits == 0 || d.decimal_point < -324) return zero;
    if (d.decimal_point >= 310) return inf;
    int32_t exp2 = 0;
    while (d.decimal_point > 0) {
        uint32_t n = (uint32_t)d.decimal_point;
        uint32_t shift = n < 19 ? powers[n] : kMaxShift;
        dec_right_shift(d, shift);
        if (d.decimal_point < -kDecPointRange) return zero;
        exp2 += (int32_t)shift;
    }
    while (d.decimal_point <= 0) {
        uint32_t shift;
        if (d.decimal_point == 0) {
            uint8_t d0 = d.digits[0];
            if (d0 >= 5) break;
            shift = (d0 == 0 || d0 == 1) ? 2 : 1;
        } else {
            uint32_t n = (uint32_t)(-d.decimal_point);
            shift = n < 19 ? powers[n] : kMaxShift;
        }
        dec_left_shift(d, shift);
        if (d.decimal_point > kDecPointRange) return inf;
        exp2 -= (int32_t)shift;
    }
    exp2 -= 1;
    while (-1022 > exp2) {
        uint32_t n = (uint32_t)(-1022 - exp2);
        if (n > kMaxShift) n = kMaxShift;
        dec_right_shift(d, n);
        exp2 += (int32_t)n;
    }
    if (exp2 + 1023 >= 0x7FF) return inf;
    dec_left_shift(d, 53);
    uint64_t mantissa = dec_round(d);
    if (mantissa >= (1ull << 53)) {
        dec_right_shift(d, 1);
        exp2 += 1;
        mantissa = dec_round(d);
        if (exp2 + 1023 >= 0x7FF) return inf;
    }
    int32_t power2 = exp2 + 1023;
    if (mantissa < (1ull << 52)) power2 -= 1;
    mantissa &= (1ull << 52) - 1;
    return BiasedFp{mantissa, power2};
}

// ---------------------------------------------------------------------------------------------
// f64::from_str.  Returns 0 = invalid, 1 = ok (*out set), 2 = needs the Decimal slow path
// (only when digit_buf == nullptr; the caller then re-runs with a buffer).
// ---------------------------------------------------------------------------------------------
template <class R>
FG_HDN int parse_f64(R& rd, uint32_t b, uint32_t e, uint8_t* digit_buf, double* out) {
    if (b >= e) return 0;
    bool negative = false;
    uint32_t c = rd.byte(b);
    if (c == '-' || c == '+') {
        negative = c == '-';
        ++b;
    }
    if (b >= e) return 0;
    // inf / infinity / nan (ASCII case-insensitive)
    if (bytes_equal_nocase(rd, b, e, "inf", 3) || bytes_equal_nocase(rd, b, e, "infinity", 8)) {
        *out = bits_to_f64(negative ? 0xFFF0000000000000ull : 0x7FF0000000000000ull);
        return 1;
    }
    if (bytes_equal_nocase(rd, b, e, "nan", 3)) {
        *out = bits_to_f64(negative ? 0xFFF8000000000000ull : 0x7FF8000000000000ull);
        return 1;
    }
    // ---- parse_number ------------------------------------------------------------------
    const uint32_t start = b;
    uint32_t p = b;
    uint64_t mantissa = 0;
    while (p < e && is_digit(rd.byte(p))) {
        mantissa = mantissa * 10u + (rd.byte(p) - '0');  // wrapping
        ++p;
    }
    int64_t n_digits = (int64_t)(p - start);
    const uint32_t int_end = p;
    int64_t n_after_dot = 0, exponent = 0;
    if (p < e && rd.byte(p) == '.') {
        ++p;
        const uint32_t before = p;
        while (p < e && is_digit(rd.byte(p))) {
            mantissa = mantissa * 10u + (rd.byte(p) - '0');
            ++p;
        }
        n_after_dot = (int64_t)(p - before);
        exponent = -n_after_dot;
    }
    n_digits += n_after_dot;
    if (n_digits == 0) return 0;
    int64_t exp_number = 0;
    if (p < e && (rd.byte(p) | 0x20u) == 'e') {
        ++p;
        bool neg_exp = false;
        if (p < e && (rd.byte(p) == '-' || rd.byte(p) == '+')) {
            neg_exp = rd.byte(p) == '-';
            ++p;
        }
        if (!(p < e && is_digit(rd.byte(p)))) return 0;
        while (p < e && is_digit(rd.byte(p))) {
            if (exp_number < 0x10000) exp_number = 10 * exp_number + (int64_t)(rd.byte(p) - '0');
            ++p;
        }
        if (neg_exp) exp_number = -exp_number;
        exponent += exp_number;
    }
    if (p != e) return 0;  // trailing characters
    bool many_digits = false;
    if (n_digits > 19) {
        n_digits -= 19;
        uint32_t q = start;
        while (q < e) {
            uint32_t ch = rd.byte(q);
            if (ch == '0') n_digits -= 1;
            else if (ch != '.') break;
            ++q;
        }
        if (n_digits > 0) {
            many_digits = true;
            mantissa = 0;
            const uint64_t kMin19 = 1000000000000000000ull;
            uint32_t s = start;
            while (mantissa < kMin19 && s < int_end) {  // integer digits
                mantissa = mantissa * 10u + (rd.byte(s) - '0');
                ++s;
            }
            if (mantissa >= kMin19) {
                exponent = (int64_t)(int_end - s);  // integer digits not consumed
            } else {
                s = int_end + 1;  // skip the '.'
                const uint32_t before = s;
                while (mantissa < kMin19 && s < e && is_digit(rd.byte(s))) {
                    mantissa = mantissa * 10u + (rd.byte(s) - '0');
                    ++s;
                }
                exponent = -(int64_t)(s - before);
            }
            exponent += exp_number;
        }
    }
    // ---- stage 1: Clinger fast path -------------------------------------------------------
    if (exponent >= -22 && exponent <= 37 && mantissa <= (1ull << 53) && !many_digits) {
        double value;
        bool ok = true;
        if (exponent <= 22) {
            value = (double)mantissa;
            if (exponent < 0) value = value / FG_POW10[-exponent];
            else value = value * FG_POW10[exponent];
        } else {
            // disguised fast path: mantissa * 10^(exponent-22) must stay <= 2^53
            uint64_t mlo, mhi;
            uint64_t p10 = 1;
            for (int64_t k = 0; k < exponent - 22; ++k) p10 *= 10u;
            mul64(mantissa, p10, &mlo, &mhi);
            if (mhi != 0 || mlo > (1ull << 53)) ok = false;
            else value = (double)mlo * FG_POW10[22];
        }
        if (ok) {
            *out = negative ? -value : value;
            return 1;
        }
    }
    // ---- stage 2: Eisel-Lemire --------------------------------------------------------------
    BiasedFp fp = compute_float(exponent, mantissa);
    if (many_digits && fp.e >= 0 && !same_fp(fp, compute_float(exponent, mantissa + 1))) fp.e = -1;
    // ---- stage 3: Decimal ---------------------------------------------------------------------
    if (fp.e < 0) {
        if (!digit_buf) return 2;
        fp = parse_long_mantissa(rd, start, e, digit_buf);
    }
    uint64_t word = fp.f | ((uint64_t)fp.e << 52) | (negative ? 0x8000000000000000ull : 0ull);
    *out = bits_to_f64(word);
    return 1;
}

// ---------------------------------------------------------------------------------------------
// serde_json 0.8 number (de.rs parse_integer / parse_decimal / parse_exponent / visit_f64_from_parts).
// [b, e) must be the complete number token candidate starting at '-' or a digit; on success
// *end = index just past the number.  kind: 2 = F64, 3 = I64, 4 = U64 (FG_T_*).
// ---------------------------------------------------------------------------------------------
FG_HD bool mul10_overflows(uint64_t a, uint64_t d) {
    return a >= 0xFFFFFFFFFFFFFFFFull / 10 && (a > 0xFFFFFFFFFFFFFFFFull / 10 || d > 0xFFFFFFFFFFFFFFFFull % 10);
}
FG_HD bool json_f64_from_parts(bool pos, uint64_t significand, int32_t exponent, uint64_t* bits) {
    double f = (double)significand;
    for (;;) {
        uint32_t a = exponent < 0 ? (uint32_t)(-(int64_t)exponent) : (uint32_t)exponent;
        if (a <= 308u) {
            if (exponent >= 0) {
                f *= FG_POW10[a];
                if (f == bits_to_f64(0x7FF0000000000000ull)) return false;  // NumberOutOfRange
            } else {
                f /= FG_POW10[a];
            }
            break;
        }
        if (f == 0.0) break;
        if (exponent >= 0) return false;
        f /= 1e308;
        exponent += 308;
    }
    *bits = f64_to_bits(pos ? f : -f);
    return true;
}
template <class R>
FG_HDN bool json_number(R& rd, uint32_t b, uint32_t e, uint32_t* end, uint32_t* kind, uint64_t* bits) {
    uint32_t p = b;
    bool pos = true;
    if (p < e && rd.byte(p) == '-') {
        pos = false;
        ++p;
    }
    if (p >= e) return false;
    uint32_t c = rd.byte(p++);
    uint64_t sig = 0;
    int32_t exponent = 0;
    bool long_int = false;
    if (c == '0') {
        if (p < e && is_digit(rd.byte(p))) return false;  // only one leading zero
    } else if (c >= '1' && c <= '9') {
        sig = c - '0';
        while (p < e && is_digit(rd.byte(p))) {
            uint64_t d = rd.byte(p) - '0';
            ++p;
            if (mul10_overflows(sig, d)) {
                long_int = true;
                exponent = 1;
                break;
            }
            sig = sig * 10u + d;
        }
        if (long_int) {
            while (p < e && is_digit(rd.byte(p))) {
                ++p;
                exponent += 1;
            }
        }
    } else {
        return false;
    }
    uint32_t nx = p < e ? rd.byte(p) : 0u;
    bool is_float = long_int;
    if (nx == '.') {
        is_float = true;
        ++p;
        bool any = false;
        while (p < e && is_digit(rd.byte(p))) {
            uint64_t d = rd.byte(p) - '0';
            ++p;
            any = true;
            if (mul10_overflows(sig, d)) {
                while (p < e && is_digit(rd.byte(p))) ++p;  // ignore all further digits
                break;
            }
            sig = sig * 10u + d;
            exponent -= 1;
        }
        if (!any) return false;
        nx = p < e ? rd.byte(p) : 0u;
    }
    if (nx == 'e' || nx == 'E') {
        is_float = true;
        ++p;
        bool pos_exp = true;
        if (p < e && rd.byte(p) == '+') ++p;
        else if (p < e && rd.byte(p) == '-') { pos_exp = false; ++p; }
        if (!(p < e && is_digit(rd.byte(p)))) return false;
        int32_t ex = (int32_t)(rd.byte(p++) - '0');
        bool exp_overflow = false;
        while (p < e && is_digit(rd.byte(p))) {
            int32_t d = (int32_t)(rd.byte(p) - '0');
            ++p;
            if (ex >= 214748364 && (ex > 214748364 || d > 7)) {
                exp_overflow = true;
                break;
            }
            ex = ex * 10 + d;
        }
        if (exp_overflow) {
            if (sig != 0 && pos_exp) return false;
            while (p < e && is_digit(rd.byte(p))) ++p;
            *end = p;
            *kind = 2;
            *bits = pos ? 0ull : 0x8000000000000000ull;
            return true;
        }
        int64_t fe = pos_exp ? (int64_t)exponent + ex : (int64_t)exponent - ex;
        if (fe > 2147483647ll) fe = 2147483647ll;
        if (fe < -2147483648ll) fe = -2147483648ll;
        exponent = (int32_t)fe;
    }
    *end = p;
    if (is_float) {
        *kind = 2;
        return json_f64_from_parts(pos, sig, exponent, bits);
    }
    if (pos) {
        *kind = 4;
        *bits = sig;
        return true;
    }
    int64_t neg = (int64_t)(0ull - sig);
    if (neg > 0) {  // magnitude above i64: becomes a float
        *kind = 2;
        *bits = f64_to_bits(-(double)sig);
    } else if (neg < 0) {
        *kind = 3;
        *bits = (uint64_t)neg;
    } else {
        *kind = 4;  // "-0" -> visit_i64(0) -> U64(0)
        *bits = 0;
    }
    return true;
}

}  // namespace num
}  // namespace fg
