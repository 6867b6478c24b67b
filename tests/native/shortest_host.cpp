// Host build of flowgger_amd/csrc/fg_shortest.hpp (what the encoder kernels run for Rust's `{}` of an f64)
// + a self-test against libstdc++'s std::to_chars (Ryu: shortest round-trip digits, closest), laid out
// the way core::fmt's digits_to_dec_str does.  Test infrastructure only.
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../flowgger_amd/csrc/fg_shortest.hpp"

namespace {
struct StrSink {  // the emitters' sink protocol: bytes, and pieces of up to sixteen (display_f64 assembles the common shapes in registers)
    static constexpr bool kCount = false;
    std::string s;
    void put(uint32_t c) { s.push_back((char)c); }
    void add(uint32_t) {}
    void put_part(uint32_t q0, uint32_t q1, uint32_t q2, uint32_t q3, uint32_t nb) {
        const uint32_t q[4] = {q0, q1, q2, q3};
        for (uint32_t i = 0; i < nb; ++i) s.push_back((char)(q[i >> 2] >> (8u * (i & 3u))));
    }
    void put16(uint32_t q0, uint32_t q1, uint32_t q2, uint32_t q3) { put_part(q0, q1, q2, q3, 16u); }
};
std::string reference_display(double v) {
    if (std::isnan(v)) return "NaN";
    std::string out;
    if (std::signbit(v)) out.push_back('-');
    v = std::fabs(v);
    if (std::isinf(v)) return out + "inf";
    if (v == 0.0) return out + "0";
    char b[64];
    auto r = std::to_chars(b, b + 64, v, std::chars_format::scientific);
    std::string t(b, r.ptr);  // d[.ddd]e[+-]xx
    size_t e = t.find('e');
    std::string digits;
    for (size_t i = 0; i < e; ++i)
        if (t[i] != '.') digits.push_back(t[i]);
    int exp10 = atoi(t.c_str() + e + 1) + 1;  // value = 0.digits * 10^exp10
    int nd = (int)digits.size();
    if (exp10 <= 0) return out + "0." + std::string((size_t)-exp10, '0') + digits;
    if (exp10 < nd) return out + digits.substr(0, (size_t)exp10) + "." + digits.substr((size_t)exp10);
    return out + digits + std::string((size_t)(exp10 - nd), '0');
}
inline uint64_t splitmix(uint64_t& x) {
    uint64_t z = (x += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
}  // namespace

extern "C" int fgs_display(double v, char* out, int cap) {
    StrSink k;
    fg::shortest::display_f64(v, k);
    if ((int)k.s.size() + 1 > cap) return -1;
    memcpy(out, k.s.c_str(), k.s.size() + 1);
    return (int)k.s.size();
}
extern "C" int fgs_reference_display(double v, char* out, int cap) {
    std::string s = reference_display(v);
    if ((int)s.size() + 1 > cap) return -1;
    memcpy(out, s.c_str(), s.size() + 1);
    return (int)s.size();
}
// mode 0: random bit patterns; 1: timestamps with 0..9 decimals; 2: m * 10^e (short decimals); 3: powers of two and
// their neighbours, subnormals; returns the number of mismatches, *bad = first offending value
extern "C" uint64_t fgs_selftest(int mode, uint64_t n, uint64_t seed, double* bad) {
    uint64_t x = seed, fails = 0;
    for (uint64_t i = 0; i < n; ++i) {
        double v;
        uint64_t r = splitmix(x);
        if (mode == 0) {
            memcpy(&v, &r, 8);
        } else if (mode == 1) {
            uint64_t secs = r % 4102444800ull;
            uint64_t r2 = splitmix(x);
            int dec = (int)(r2 % 10);
            uint64_t p = 1;
            for (int k = 0; k < dec; ++k) p *= 10;
            v = (double)secs + (double)((r2 >> 8) % p) / (double)p;
            if (r2 & 0x80) v = -v;
        } else if (mode == 2) {
            uint64_t r2 = splitmix(x);
            int nd = 1 + (int)(r2 % 17);
            uint64_t p = 1;
            for (int k = 0; k < nd; ++k) p *= 10;
            double m = (double)(r % p);
            int e = (int)((r2 >> 8) % 600) - 300;
            v = m * std::pow(10.0, e);
        } else {
            uint64_t r2 = splitmix(x);
            uint64_t bexp = r % 2047;
            uint64_t frac = (r2 & 1) ? 0 : ((r2 & 2) ? 0x000FFFFFFFFFFFFFull : (r2 >> 12));
            if (r2 & 4) frac ^= (r2 >> 60);
            uint64_t b = (bexp << 52) | (frac & 0x000FFFFFFFFFFFFFull);
            memcpy(&v, &b, 8);
        }
        StrSink k;
        fg::shortest::display_f64(v, k);
        if (k.s != reference_display(v)) {
            if (!fails && bad) *bad = v;
            ++fails;
        }
    }
    return fails;
}
