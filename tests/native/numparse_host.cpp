// Host build of flowgger_amd/csrc/fg_numparse.hpp for CPU fuzzing against the oracle (strtod).
// Test infrastructure: lets the exact code the kernels run be checked here without a GPU.
#include <cstdint>
#include <cstring>

#include "../../flowgger_amd/csrc/fg_numparse.hpp"

struct PtrReader {
    const uint8_t* p;
    uint32_t byte(uint32_t i) const { return p[i]; }
};

extern "C" int fgn_parse_f64(const uint8_t* s, uint32_t len, int allow_slow, double* out) {
    PtrReader rd{s};
    uint8_t buf[fg::num::kDecMaxDigits];
    return fg::num::parse_f64(rd, 0, len, allow_slow ? buf : nullptr, out);
}
extern "C" int fgn_json_number(const uint8_t* s, uint32_t len, uint32_t* end, uint32_t* kind, uint64_t* bits) {
    PtrReader rd{s};
    return fg::num::json_number(rd, 0, len, end, kind, bits) ? 1 : 0;
}
extern "C" int fgn_parse_u64(const uint8_t* s, uint32_t len, uint64_t max, uint64_t* out) {
    PtrReader rd{s};
    return fg::num::parse_unsigned(rd, 0, len, max, out) ? 1 : 0;
}
extern "C" int fgn_parse_i64(const uint8_t* s, uint32_t len, int64_t* out) {
    PtrReader rd{s};
    return fg::num::parse_i64(rd, 0, len, out) ? 1 : 0;
}
