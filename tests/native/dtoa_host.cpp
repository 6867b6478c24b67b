// Host build of flowgger_amd/csrc/fg_dtoa.hpp (the exact f64 -> text code the GELF encoder kernel runs),
// so that it can be compared with the oracle's independent restatement on the CPU.  Test infrastructure only.
#include <cstdint>

#include "../../flowgger_amd/csrc/fg_dtoa.hpp"

extern "C" int fgd_write(double v, char* out) { return fg::dtoa::write(v, out); }
extern "C" uint64_t fgd_write_batch(const double* v, uint64_t n, char* out /* 32 bytes per value, NUL terminated */) {
    for (uint64_t i = 0; i < n; ++i) {
        int k = fg::dtoa::write(v[i], out + 32 * i);
        out[32 * i + k] = 0;
    }
    return n;
}
