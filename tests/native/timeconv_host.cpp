// Host build of flowgger_amd/csrc/fg_timeconv.hpp: lets the exact timestamp arithmetic the
// kernels run be checked on the CPU (against the hardware divider and Python big integers).
// Test infrastructure only.
#include <cstdint>

#include "../../flowgger_amd/csrc/fg_timeconv.hpp"

extern "C" void fgt_div1e9_batch(const double* x, double* out, uint64_t n) {
    for (uint64_t i = 0; i < n; ++i) out[i] = fg::div_by_1e9(x[i]);
}
// out = our 3-op quotient; returns the number of mismatches against x / 1e9 (volatile divisor so
// that the compiler cannot turn the reference into a reciprocal multiply)
extern "C" uint64_t fgt_div1e9_sweep(uint64_t seed, uint64_t n, int mode, double* first_bad) {
    volatile double y = 1e9;
    uint64_t s = seed, bad = 0;
    for (uint64_t i = 0; i < n; ++i) {
        s += 0x9E3779B97F4A7C15ull;
        uint64_t z = s;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        uint64_t v;
        switch (mode) {
            case 0: v = z; break;                                   // any 64-bit integer
            case 1: v = z >> (z & 63); break;                       // all magnitudes
            case 2: v = (z % 2500000000ull) * 1000000000ull + (z >> 34) % 1000000000ull; break;  // plausible stamps
            default: v = ((z >> 20) % (1ull << 34)) * 1000000000ull + (z & 1023); break;        // tiny fractions
        }
        double x = (double)v;
        if (mode & 4) x = -x;
        double a = fg::div_by_1e9(x), b = x / y;
        if (!(a == b)) {
            if (bad == 0 && first_bad) *first_bad = x;
            ++bad;
        }
    }
    return bad;
}
extern "C" void fgt_unix_nanos_batch(const int64_t* secs, const uint32_t* nano, double* out, uint64_t n) {
    for (uint64_t i = 0; i < n; ++i) out[i] = fg::unix_nanos_to_f64(secs[i], nano[i]);
}
// parts: y mo d h mi s nano sign oh om (10 ints per row); rc: 1 ok / 0 invalid
extern "C" void fgt_datetime_batch(const int32_t* parts, uint64_t n, int allow_leap, int32_t* rc, double* out,
                                   int32_t* rc_fast, double* out_fast) {
    for (uint64_t i = 0; i < n; ++i) {
        const int32_t* q = parts + 10 * i;
        fg::DateTimeParts p{q[0], q[1], q[2], q[3], q[4], q[5], (uint32_t)q[6], q[7], q[8], q[9]};
        out[i] = 0.0;
        out_fast[i] = 0.0;
        rc[i] = fg::datetime_to_unix(p, allow_leap != 0, &out[i]) ? 1 : 0;
        rc_fast[i] = fg::datetime_to_unix_fast(p, &out_fast[i]);
    }
}
