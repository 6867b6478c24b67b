// fg_dtoa.hpp -- f64 -> text exactly as the `dtoa` crate (0.2 / 0.4, what serde_json 0.8 uses for
// Value::F64) prints it: Florian Loitsch's Grisu2 as implemented in rapidjson (DiyFp, the 87-entry
// cached-power table, DigitGen with GrisuRound) + rapidjson's Prettify (decimal notation for
// 1e-6 <= v < 1e21, "d.ddde[-]x" otherwise, ".0" after integral values).  Host + device; checked
// on the CPU against the oracle's independent restatement and Python's float parser
// (tests/test_encoder_cpu.py).
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define FGD_HD __host__ __device__ __forceinline__
#else
#define FGD_HD inline
#endif

namespace fg {
namespace dtoa {

struct DiyFp {
    uint64_t f;
    int e;
};

#if defined(__HIP_DEVICE_COMPILE__)
#define FGD_TABLE static __device__ const
#else
#define FGD_TABLE static const
#endif
FGD_TABLE uint64_t kCachedF[87] = {
    0xfa8fd5a0081c0288ull,
    0xbaaee17fa23ebf76ull,
    0x8b16fb203055ac76ull,
    0xcf42894a5dce35eaull,
    0x9a6bb0aa55653b2dull,
    0xe61acf033d1a45dfull,
    0xab70fe17c79ac6caull,
    0xff77b1fcbebcdc4full,
    0xbe5691ef416bd60cull,
    0x8dd01fad907ffc3cull,
    0xd3515c2831559a83ull,
    0x9d71ac8fada6c9b5ull,
    0xea9c227723ee8bcbull,
    0xaecc49914078536dull,
    0x823c12795db6ce57ull,
    0xc21094364dfb5637ull,
    0x9096ea6f3848984full,
    0xd77485cb25823ac7ull,
    0xa086cfcd97bf97f4ull,
    0xef340a98172aace5ull,
    0xb23867fb2a35b28eull,
    0x84c8d4dfd2c63f3bull,
    0xc5dd44271ad3cdbaull,
    0x936b9fcebb25c996ull,
    0xdbac6c247d62a584ull,
    0xa3ab66580d5fdaf6ull,
    0xf3e2f893dec3f126ull,
    0xb5b5ada8aaff80b8ull,
    0x87625f056c7c4a8bull,
    0xc9bcff6034c13053ull,
    0x964e858c91ba2655ull,
    0xdff9772470297ebdull,
    0xa6dfbd9fb8e5b88full,
    0xf8a95fcf88747d94ull,
    0xb94470938fa89bcfull,
    0x8a08f0f8bf0f156bull,
    0xcdb02555653131b6ull,
    0x993fe2c6d07b7facull,
    0xe45c10c42a2b3b06ull,
    0xaa242499697392d3ull,
    0xfd87b5f28300ca0eull,
    0xbce5086492111aebull,
    0x8cbccc096f5088ccull,
    0xd1b71758e219652cull,
    0x9c40000000000000ull,
    0xe8d4a51000000000ull,
    0xad78ebc5ac620000ull,
    0x813f3978f8940984ull,
    0xc097ce7bc90715b3ull,
    0x8f7e32ce7bea5c70ull,
    0xd5d238a4abe98068ull,
    0x9f4f2726179a2245ull,
    0xed63a231d4c4fb27ull,
    0xb0de65388cc8ada8ull,
    0x83c7088e1aab65dbull,
    0xc45d1df942711d9aull,
    0x924d692ca61be758ull,
    0xda01ee641a708deaull,
    0xa26da3999aef774aull,
    0xf209787bb47d6b85ull,
    0xb454e4a179dd1877ull,
    0x865b86925b9bc5c2ull,
    0xc83553c5c8965d3dull,
    0x952ab45cfa97a0b3ull,
    0xde469fbd99a05fe3ull,
    0xa59bc234db398c25ull,
    0xf6c69a72a3989f5cull,
    0xb7dcbf5354e9beceull,
    0x88fcf317f22241e2ull,
    0xcc20ce9bd35c78a5ull,
    0x98165af37b2153dfull,
    0xe2a0b5dc971f303aull,
    0xa8d9d1535ce3b396ull,
    0xfb9b7cd9a4a7443cull,
    0xbb764c4ca7a44410ull,
    0x8bab8eefb6409c1aull,
    0xd01fef10a657842cull,
    0x9b10a4e5e9913129ull,
    0xe7109bfba19c0c9dull,
    0xac2820d9623bf429ull,
    0x80444b5e7aa7cf85ull,
    0xbf21e44003acdd2dull,
    0x8e679c2f5e44ff8full,
    0xd433179d9c8cb841ull,
    0x9e19db92b4e31ba9ull,
    0xeb96bf6ebadf77d9ull,
    0xaf87023b9bf0ee6bull
};
FGD_TABLE int16_t kCachedE[87] = {
    -1220,
    -1193,
    -1166,
    -1140,
    -1113,
    -1087,
    -1060,
    -1034,
    -1007,
    -980,
    -954,
    -927,
    -901,
    -874,
    -847,
    -821,
    -794,
    -768,
    -741,
    -715,
    -688,
    -661,
    -635,
    -608,
    -582,
    -555,
    -529,
    -502,
    -475,
    -449,
    -422,
    -396,
    -369,
    -343,
    -316,
    -289,
    -263,
    -236,
    -210,
    -183,
    -157,
    -130,
    -103,
    -77,
    -50,
    -24,
    3,
    30,
    56,
    83,
    109,
    136,
    162,
    189,
    216,
    242,
    269,
    295,
    322,
    348,
    375,
    402,
    428,
    455,
    481,
    508,
    534,
    561,
    588,
    614,
    641,
    667,
    694,
    720,
    747,
    774,
    800,
    827,
    853,
    880,
    907,
    933,
    960,
    986,
    1013,
    1039,
    1066
};
FGD_TABLE uint32_t kPow10[10] = {1, 10, 100, 1000, 10000, 100000, 1000000, 10000000, 100000000, 1000000000};
#undef FGD_TABLE

FGD_HD uint64_t mulhi_round(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint64_t h = __umul64hi(a, b), l = a * b;
#else
    const unsigned __int128 p = (unsigned __int128)a * b;
    const uint64_t h = (uint64_t)(p >> 64), l = (uint64_t)p;
#endif
    return h + (l >> 63);
}
FGD_HD DiyFp mul(DiyFp a, DiyFp b) { return DiyFp{mulhi_round(a.f, b.f), a.e + b.e + 64}; }

// The digits live in a register as a decimal integer (at most 17 digits fit a u64), never in a char buffer: a
// dynamically indexed local array is scratch memory on the GPU, i.e. one dependent memory round trip per digit.
struct Digits {
    uint64_t v = 0;  // the digits read as one integer
    int len = 0;     // how many (leading zeros are never generated)
    FGD_HD void push(uint32_t d) {
        if (d || len) {
            v = v * 10u + d;
            ++len;
        }
    }
};
FGD_HD void grisu_round(Digits& dg, uint64_t delta, uint64_t rest, uint64_t ten_kappa, uint64_t wp_w) {
    while (rest < wp_w && delta - rest >= ten_kappa && (rest + ten_kappa < wp_w || wp_w - rest > rest + ten_kappa - wp_w)) {
        --dg.v;  // --buffer[len - 1]: the last digit (Grisu never takes it below '0')
        rest += ten_kappa;
    }
}
FGD_HD int count_digits32(uint32_t n) {
    int c = 1;
    c += n >= 10u;
    c += n >= 100u;
    c += n >= 1000u;
    c += n >= 10000u;
    c += n >= 100000u;
    c += n >= 1000000u;
    c += n >= 10000000u;
    c += n >= 100000000u;
    c += n >= 1000000000u;
    return c;
}
FGD_HD void digit_gen(DiyFp W, DiyFp Mp, uint64_t delta, Digits& dg, int* K) {
    const int sh = -Mp.e;
    const uint64_t one_f = 1ull << sh;
    const uint64_t wp_w = Mp.f - W.f;
    uint32_t p1 = (uint32_t)(Mp.f >> sh);
    uint64_t p2 = Mp.f & (one_f - 1);
    int kappa = count_digits32(p1);
    while (kappa > 0) {
        const uint32_t div = kPow10[kappa - 1];
        const uint32_t d = p1 / div;
        p1 -= d * div;
        dg.push(d);
        --kappa;
        const uint64_t tmp = ((uint64_t)p1 << sh) + p2;
        if (tmp <= delta) {
            *K += kappa;
            grisu_round(dg, delta, tmp, (uint64_t)kPow10[kappa] << sh, wp_w);
            return;
        }
    }
    for (;;) {
        p2 *= 10;
        delta *= 10;
        const uint32_t d = (uint32_t)(p2 >> sh) & 0xFFu;
        dg.push(d);
        p2 &= one_f - 1;
        --kappa;
        if (p2 < delta) {
            *K += kappa;
            const int index = -kappa;
            grisu_round(dg, delta, p2, one_f, wp_w * (index < 9 ? kPow10[index] : 0u));
            return;
        }
    }
}
FGD_HD void grisu2(double value, Digits& dg, int* K) {
    uint64_t u;
    memcpy(&u, &value, 8);
    const int biased = (int)((u >> 52) & 0x7FFu);
    const uint64_t frac = u & ((1ull << 52) - 1);
    const DiyFp v = biased ? DiyFp{frac + (1ull << 52), biased - 1075} : DiyFp{frac, -1074};
    DiyFp pl{(v.f << 1) + 1, v.e - 1};
    while (!(pl.f & (1ull << 53))) {
        pl.f <<= 1;
        --pl.e;
    }
    pl.f <<= 10;
    pl.e -= 10;
    DiyFp mi = (v.f == (1ull << 52)) ? DiyFp{(v.f << 2) - 1, v.e - 2} : DiyFp{(v.f << 1) - 1, v.e - 1};
    mi.f <<= mi.e - pl.e;
    mi.e = pl.e;
    // cached power for pl.e
    const double dk = (-61 - pl.e) * 0.30102999566398114 + 347;
    int k = (int)dk;
    if (dk - k > 0.0) ++k;
    const unsigned index = (unsigned)((k >> 3) + 1);
    *K = -(-348 + (int)(index << 3));
    const DiyFp c_mk{kCachedF[index], (int)kCachedE[index]};
    const int s = __builtin_clzll(v.f);
    const DiyFp W = mul(DiyFp{v.f << s, v.e - s}, c_mk);
    DiyFp Wp = mul(pl, c_mk);
    DiyFp Wm = mul(mi, c_mk);
    ++Wm.f;
    --Wp.f;
    digit_gen(W, Wp, Wp.f - Wm.f, dg, K);
}

// The generated digits as 17 BCD nibbles (digit 0 first; digits past len are zero) for reading digit i with shifts.
struct Bcd17 {
    uint32_t top;   // digit 0
    uint32_t a, b;  // digits 1..8 and 9..16, eight nibbles each, the first digit in the highest nibble
    FGD_HD static uint32_t bcd8(uint32_t x) {  // x < 10^8
        uint32_t r = 0;
        r |= (x / 10000000u) << 28;
        r |= (x / 1000000u % 10u) << 24;
        r |= (x / 100000u % 10u) << 20;
        r |= (x / 10000u % 10u) << 16;
        r |= (x / 1000u % 10u) << 12;
        r |= (x / 100u % 10u) << 8;
        r |= (x / 10u % 10u) << 4;
        r |= x % 10u;
        return r;
    }
    FGD_HD explicit Bcd17(const Digits& dg) {
        uint64_t n = dg.v;
        for (int j = dg.len; j < 17; ++j) n *= 10u;  // left-align to 17 digits
        top = (uint32_t)(n / 10000000000000000ull);
        const uint64_t rest = n % 10000000000000000ull;
        a = bcd8((uint32_t)(rest / 100000000ull));
        b = bcd8((uint32_t)(rest % 100000000ull));
    }
    FGD_HD uint32_t digit(int i) const {  // 0 <= i < 17
        if (i == 0) return top;
        const uint32_t w = i <= 8 ? a : b;
        const int j = i <= 8 ? i - 1 : i - 9;
        return (w >> (4 * (7 - j))) & 15u;
    }
};

// The digits dg * 10^k as rapidjson's Prettify prints them, streamed into sink.put(char code): at most 25 characters.
template <class Sink>
FGD_HD void stream_digits(const Digits& dg, int k, Sink& sink) {
    const int len = dg.len;
    const Bcd17 bcd(dg);
    const int kk = len + k;  // 10^(kk-1) <= v < 10^kk
    // ONE stream: `lead` zeros ("0." + zeros for 1234e-6 -> 0.001234), the digits with a '.'
    // after digit `dot` (0: none), `trail` zeros and ".0" (1234e7 -> 12340000000.0), or an exponent
    int dot = 0, trail = 0, lead = 0;
    bool exp = false, point_zero = false;
    if (0 <= k && kk <= 21) {
        trail = k;
        point_zero = true;
    } else if (0 < kk && kk <= 21) {
        dot = kk;
    } else if (-6 < kk && kk <= 0) {
        lead = -kk;
        sink.put((uint32_t)'0');
        sink.put((uint32_t)'.');
    } else {
        exp = true;
        dot = len > 1 ? 1 : 0;  // 1e30 / 1.234e33
    }
    for (int i = 0; i < lead; ++i) sink.put((uint32_t)'0');
    for (int i = 0; i < len; ++i) {
        if (dot && i == dot) sink.put((uint32_t)'.');
        sink.put((uint32_t)'0' + bcd.digit(i));
    }
    for (int i = 0; i < trail; ++i) sink.put((uint32_t)'0');
    if (point_zero) {
        sink.put((uint32_t)'.');
        sink.put((uint32_t)'0');
    }
    if (exp) {
        sink.put((uint32_t)'e');
        int K = kk - 1;
        if (K < 0) {
            sink.put((uint32_t)'-');
            K = -K;
        }
        if (K >= 100) sink.put((uint32_t)'0' + (uint32_t)(K / 100));
        if (K >= 10) sink.put((uint32_t)'0' + (uint32_t)(K / 10 % 10));
        sink.put((uint32_t)'0' + (uint32_t)(K % 10));
    }
}
// the sign and the zeros; false = the text is complete
template <class Sink>
FGD_HD bool write_sign_zero(double& value, Sink& sink) {
    uint64_t u;
    memcpy(&u, &value, 8);
    if (u >> 63) {
        sink.put((uint32_t)'-');
        value = -value;
    }
    if ((u << 1) == 0) {  // +-0.0
        sink.put((uint32_t)'0');
        sink.put((uint32_t)'.');
        sink.put((uint32_t)'0');
        return false;
    }
    return true;
}
// value must be finite.  Streams at most 26 characters into sink.put(char code).
template <class Sink>
FGD_HD void write_to(double value, Sink& sink) {
    if (!write_sign_zero(value, sink)) return;
    Digits dg;
    int k = 0;
    grisu2(value, dg, &k);
    stream_digits(dg, k, sink);
}

// four BCD digits (the low 16 bits of y, the first digit in the highest nibble) as four ASCII bytes, the first digit in byte 0
FGD_HD uint32_t ascii4(uint32_t y) {
    uint32_t z = (y | (y << 8)) & 0x00FF00FFu;
    z = (z | (z << 4)) & 0x0F0F0F0Fu;  // (the first digit in byte 3)
    return __builtin_bswap32(z) + 0x30303030u;
}
// The 17 digits of `bcd` as text with a '.' in front of digit `dot` (1 .. 16; >= 20: none), cut to L characters (<= 18), as one piece
// or two: put16 / put_part of the sink.  (The digits beyond the generated ones are zeros: trailing zeros of an integer come for free.)
template <class Sink>
FGD_HD void put_text17(const Bcd17& bcd, uint32_t dot, uint32_t L, Sink& sink) {
    const uint32_t u0 = ascii4(bcd.a >> 16), u1 = ascii4(bcd.a & 0xFFFFu), u2 = ascii4(bcd.b >> 16), u3 = ascii4(bcd.b & 0xFFFFu);
    const uint32_t S[5] = {((uint32_t)'0' + bcd.top) | u0 << 8, u0 >> 24 | u1 << 8, u1 >> 24 | u2 << 8, u2 >> 24 | u3 << 8, u3 >> 24};
    uint32_t R[5];
#ifdef __HIP_DEVICE_COMPILE__
#pragma unroll
#endif
    for (uint32_t j = 0; j < 5u; ++j) {
        const uint32_t shifted = (j ? S[j - 1u] >> 24 : 0u) | S[j] << 8;        // the string moved up by the '.'
        const uint32_t v = dot > 4u * j ? dot - 4u * j : 0u;                    // this dword's bytes in front of the '.'
        const uint32_t m = v >= 4u ? 0xFFFFFFFFu : (1u << (8u * v)) - 1u;
        uint32_t r = (S[j] & m) | (shifted & ~m);
        if (v < 4u && dot >= 4u * j) r = (r & ~(0xFFu << (8u * v))) | (uint32_t)'.' << (8u * v);
        const uint32_t keep = L > 4u * j ? L - 4u * j : 0u;                     // ... and of the text at all
        R[j] = keep >= 4u ? r : (r & ((1u << (8u * keep)) - 1u));
    }
    if (L >= 16u) {
        sink.put16(R[0], R[1], R[2], R[3]);
        if (L > 16u) sink.put_part(R[4], 0u, 0u, 0u, L - 16u);
    } else {
        sink.put_part(R[0], R[1], R[2], R[3], L);
    }
}
// The same text for a sink that takes PIECES (fg_emit.hpp: put16 / put_part / add, kCount): a timestamp -- seconds with a fraction, or
// whole seconds: the decimal point falls inside the first 17 positions -- is assembled in registers and handed over as one piece or
// two instead of seventeen to twenty put() calls (what a put costs the write pass: PackSink::put_part).  Everything else streams.
template <class Sink>
FGD_HD void write_pieces(double value, Sink& sink) {
    if (!write_sign_zero(value, sink)) return;
    Digits dg;
    int k = 0;
    grisu2(value, dg, &k);
    const int len = dg.len, kk = len + k;
    if (!(0 < kk && kk <= 16)) {
        stream_digits(dg, k, sink);
        return;
    }
    // digits [0, kk) '.' digits [kk, len)   (k < 0: len + 1 characters)   or   digits, k zeros, ".0"   (k >= 0: kk + 2 characters) --
    // the 17 BCD digits are zero beyond len, so the string S of their ASCII codes holds the trailing zeros already
    const uint32_t L = (uint32_t)(k < 0 ? len + 1 : kk + 2), dot = (uint32_t)kk;
    if (Sink::kCount) {
        sink.add(L);
        return;
    }
    put_text17(Bcd17(dg), dot, L, sink);
}
// Buffer form (host tests): writes at most 26 characters into out, returns the count.  Through write_pieces, so that the tests of
// this function cover the assembled form as well as the streamed one.
struct BufSink {
    static constexpr bool kCount = false;
    char* p;
    int n;
    FGD_HD void put(uint32_t c) { p[n++] = (char)c; }
    FGD_HD void add(uint32_t) {}
    FGD_HD void put_part(uint32_t q0, uint32_t q1, uint32_t q2, uint32_t q3, uint32_t nb) {
        const uint32_t q[4] = {q0, q1, q2, q3};
        for (uint32_t i = 0; i < nb; ++i) p[n++] = (char)(q[i >> 2] >> (8u * (i & 3u)));
    }
    FGD_HD void put16(uint32_t q0, uint32_t q1, uint32_t q2, uint32_t q3) { put_part(q0, q1, q2, q3, 16u); }
};
FGD_HD int write(double value, char* out) {
    BufSink bs{out, 0};
    write_pieces(value, bs);
    return bs.n;
}

}  // namespace dtoa
}  // namespace fg
