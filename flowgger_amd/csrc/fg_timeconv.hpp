// fg_timeconv.hpp -- calendar -> Unix timestamp arithmetic with the reference's exact result,
// usable from host (unit-tested on the CPU against true IEEE division / Python big ints) and
// device code.
//
//   PreciseTimestamp::from_offset_datetime (reference: src/flowgger/utils/mod.rs:23-28):
//       ts = (unix_timestamp_nanos() as i128) as f64 / 1e9
//   i.e. integer nanoseconds -> f64 (round to nearest even) -> ONE correctly rounded division.
//   time 0.3: Date::from_calendar_date, Time::from_hms_nano, UtcOffset::from_hms,
//   OffsetDateTime::unix_timestamp_nanos.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define FGT_HD __host__ __device__ __forceinline__
#else
#define FGT_HD inline
#endif

namespace fg {

// x / 1e9, correctly rounded, for x = an INTEGER of magnitude < 2^64 held in an f64 -- in three
// FMA-class operations instead of the ~15-operation IEEE division expansion (no f64 divider on
// the GPU).  q0 = RN(x*z) with z = RN(1e-9) is within 1.56 ulp of x/1e9; r = x - q0*1e9 is
// exactly representable (a multiple of 2^(E-43) below 2^(E-20)), so the FMA computes it exactly;
// q0 + r*z differs from x/1e9 by < 2^-53 ulp, while x/1e9 -- x an integer, 1e9 = 2^9 * 5^9 --
// is never closer than 2^-22 ulp to a rounding boundary and never exactly on one (a midpoint
// times 1e9 needs 75 significant bits).  Hence RN(q0 + r*z) == RN(x / 1e9).  Checked against the
// hardware divider on 10^8 values in tests/test_timeconv_cpu.py.
FGT_HD double div_by_1e9(double x) {
    const double z = 1e-9;
    double q0 = x * z;
    double r = __builtin_fma(-q0, 1e9, x);
    return __builtin_fma(r, z, q0);
}

FGT_HD uint64_t umul64hi_(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}

// (secs * 1e9 + nano) as i128 -> f64 (RNE) -> / 1e9, without __int128 runtime support.
FGT_HD double unix_nanos_to_f64(int64_t secs, uint32_t nano) {
    // total = secs*1e9 + nano ; nano in [0, 1e9).  Work on the magnitude.
    bool neg = secs < 0;
    uint64_t lo, hi;
    if (!neg) {
        uint64_t a = (uint64_t)secs;
        lo = a * 1000000000ull;
        hi = umul64hi_(a, 1000000000ull);
        uint64_t l2 = lo + nano;
        hi += l2 < lo;
        lo = l2;
    } else {
        // |total| = (-secs)*1e9 - nano   (secs <= -1 so this is > 0)
        uint64_t a = (uint64_t)(-secs);
        lo = a * 1000000000ull;
        hi = umul64hi_(a, 1000000000ull);
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
        int s = 64 - __builtin_clzll(hi);  // 1..64; here hi < 2^5
        uint64_t m = (hi << (64 - s)) | (lo >> s);
        uint64_t lost = lo & ((1ull << s) - 1ull);
        m |= (lost != 0);
        mag = __builtin_ldexp((double)m, s);  // exact scaling
        double f = neg ? -mag : mag;
        return f / 1e9;  // beyond 2^64 ns (years outside 1385..2554): plain IEEE division
    }
    return div_by_1e9(neg ? -mag : mag);
}

FGT_HD bool is_leap_year(int y) { return (y % 4 == 0) && (y % 100 != 0 || y % 400 == 0); }
FGT_HD int days_in_month(int y, int m) {
    // 31 28 31 30 31 30 31 31 30 31 30 31 packed 2 bits each above 28
    int d = 28 + ((0xEEFBB3 >> ((m - 1) * 2)) & 3);  // Jan..Dec extra days: 3 0 3 2 3 2 3 3 2 3 2 3
    return (m == 2 && is_leap_year(y)) ? 29 : d;
}
FGT_HD int64_t days_from_civil(int y, int m, int d) {
    y -= m <= 2;
    int era = (y >= 0 ? y : y - 399) / 400;
    int yoe = y - era * 400;
    int doy = (153 * (m + (m > 2 ? -3 : 9)) + 2) / 5 + d - 1;
    int doe = yoe * 365 + yoe / 4 - yoe / 100 + doy;
    return (int64_t)era * 146097 + doe - 719468;
}
FGT_HD void civil_from_days(int64_t z, int* y, int* m, int* d) {
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

struct DateTimeParts {
    int year, month, day, hour, minute, second;
    uint32_t nano;
    int off_sign, off_h, off_m;
};
// Validation + conversion; allow_leap = the Rfc3339 parser's second==60 stand-in.
FGT_HD bool datetime_to_unix(const DateTimeParts& p, bool allow_leap, double* out) {
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

// Branch-free form for the common shape: years 0..9999 from four digits, no leap second.
// Returns 1 = converted, 0 = invalid field, 2 = outside the fast domain (second == 60, or a
// result before 1970 / beyond 2^34 s) -> the caller uses datetime_to_unix.
FGT_HD int datetime_to_unix_fast(const DateTimeParts& p, double* out) {
    const uint32_t mo = (uint32_t)p.month, dy = (uint32_t)p.day;
    const bool valid = (mo - 1u) <= 11u && (dy - 1u) < (uint32_t)days_in_month(p.year, (int)(mo - 1u <= 11u ? mo : 1u)) &&
                       (uint32_t)p.hour <= 23u && (uint32_t)p.minute <= 59u && (uint32_t)p.second <= 59u &&
                       (uint32_t)p.off_h <= 25u && (uint32_t)p.off_m <= 59u;
    // days_from_civil for year >= 0 (era arithmetic without the negative branch)
    const int m = (int)(mo - 1u <= 11u ? mo : 1u);
    const int y = p.year - (m <= 2);
    const int era = (y + 400) / 400 - 1;             // y >= -1
    const int yoe = y - era * 400;
    const int doy = (153 * (m + (m > 2 ? -3 : 9)) + 2) / 5 + (int)dy - 1;
    const int doe = yoe * 365 + yoe / 4 - yoe / 100 + doy;
    const int64_t days = (int64_t)era * 146097 + doe - 719468;
    const int off = p.off_sign * (p.off_h * 3600 + p.off_m * 60);
    const int64_t secs = days * 86400 + (p.hour * 3600 + p.minute * 60 + p.second - off);
    const bool in_domain = p.second != 60 && secs >= 0 && secs < (1ll << 34);
    const uint64_t nanos = (uint64_t)secs * 1000000000ull + p.nano;  // < 2^64 inside the domain
    *out = div_by_1e9((double)nanos);
    if (p.second == 60) return 2;
    if (!valid) return 0;
    return in_domain ? 1 : 2;
}

}  // namespace fg
