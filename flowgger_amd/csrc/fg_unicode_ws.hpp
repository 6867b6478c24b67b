// fg_unicode_ws.hpp -- char::is_whitespace (Unicode White_Space) on UTF-8 bytes served by a byte reader; host + device.
// Shared by the RFC3164 parser (str::split_whitespace, rfc3164_decoder.rs:58) and the encoders' re-join of the
// RFC3164 message (fg_emit.hpp); kept apart so that parser changes do not rebuild the (slow to compile) encoders.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define FG3_HD __host__ __device__ __forceinline__
#else
#define FG3_HD inline
#endif

namespace fg {
namespace r3164 {

// byte length of the Unicode White_Space character starting at rd[i] (0: not whitespace); char::is_whitespace
template <class R>
FG3_HD uint32_t ws_at(R& rd, uint32_t i, uint32_t end) {
    const uint32_t c = rd.byte(i);
    if (c < 0x80u) return (c == 32u || (c - 9u) <= 4u) ? 1u : 0u;
    if (c == 0xC2u) {
        if (i + 1 >= end) return 0;
        const uint32_t d = rd.byte(i + 1);
        return (d == 0x85u || d == 0xA0u) ? 2u : 0u;
    }
    if (c == 0xE1u || c == 0xE2u || c == 0xE3u) {
        if (i + 2 >= end) return 0;
        const uint32_t b1 = rd.byte(i + 1), b2 = rd.byte(i + 2);
        if (c == 0xE2u) {
            if (b1 == 0x80u) return ((b2 - 0x80u) <= 0x0Au || b2 == 0xA8u || b2 == 0xA9u || b2 == 0xAFu) ? 3u : 0u;
            return (b1 == 0x81u && b2 == 0x9Fu) ? 3u : 0u;
        }
        if (c == 0xE1u) return (b1 == 0x9Au && b2 == 0x80u) ? 3u : 0u;
        return (b1 == 0x80u && b2 == 0x80u) ? 3u : 0u;
    }
    return 0;
}
}  // namespace r3164
}  // namespace fg
