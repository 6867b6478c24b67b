// fg_emit.hpp -- the five Encoder::encode implementations + the three Merger::frame implementations, as
// per-record emitters that read a decode-table row (spans into the packed line bytes + the entry slice)
// and push the output bytes into a sink -- the Record is never materialised.
//
//   GelfEmitter         encoder/gelf_encoder.rs:59-115 (+ serde_json 0.8 serialisation of the BTreeMap)
//   LtsvEmitter         encoder/ltsv_encoder.rs:33-125
//   Rfc5424Emitter      encoder/rfc5424_encoder.rs:28-93  (+ impl Display for StructuredData, record.rs:42-67)
//   Rfc3164Emitter      encoder/rfc3164_encoder.rs:28-101
//   PassthroughEmitter  encoder/passthrough_encoder.rs:24-50
//   frame               merger/{line,nul,syslen}_merger.rs
//
// A Record field is a DECODED view of a span: raw bytes (RFC5424 / LTSV sources), unescape_sd_value
// (rfc5424_decoder.rs:105-125) for flagged RFC5424 SD values, JSON string unescaping (serde_json 0.8) for flagged
// spans of GELF-decoded records.  `for_each_decoded` streams that view byte by byte; every emitter is written
// against it, so the three decoders x five encoders all share one code path.
//
// Host + device (no HIP dependency): the kernels in fg_encode.hip instantiate the emitters with a GlobalReader and
// a count / write sink; tests/native/emit_host.cpp instantiates the same code on the CPU against the oracle.
#pragma once
#include <stdint.h>
#include <string.h>

#include "fg_dtoa.hpp"
#include "fg_unicode_ws.hpp"
#include "fg_shortest.hpp"
#include "fg_tables_view.hpp"
#include "fg_timeconv.hpp"

#if defined(__HIPCC__)
#define FGE_HD __host__ __device__ __forceinline__
#define FGE_HD_NOINLINE __host__ __device__
#else
#define FGE_HD inline
#define FGE_HD_NOINLINE inline
#endif

namespace fg {

// static key list entry of the GELF encoder (host-built, sorted by key): what the value is
enum : uint32_t { SK_APP = 0, SK_FULL = 1, SK_HOST = 2, SK_LEVEL = 3, SK_PROC = 4, SK_SDID = 5, SK_SHORT = 6, SK_TS = 7, SK_VERSION = 8, SK_EXTRA = 9 };
struct StaticKey {
    uint32_t key_off, key_len;    // the raw key bytes (for the merge with the record's own keys), into the blob
    uint32_t kind;                // SK_*
    // The member text up to the value's first own byte, TWICE (both 4-byte aligned in the blob, text_len bytes each): at text_off
    // behind a ',' -- `,"key":` with the key JSON-escaped --, at text1_off behind the '{' of an object it opens.  The kinds whose value
    // is always a string carry its opening quote (`,"host":"`), SK_VERSION and SK_EXTRA the whole member (`,"version":"1.1"`,
    // `,"key":"value"`): every byte that does not depend on the record is part of ONE piece (what a piece costs: PackSink::put_part).
    uint32_t text_off, text1_off, text_len;
};
struct EncCfg {
    const uint8_t* blob;       // GELF keys + extra values, LTSV suffixes, the LTSV extras text, the prepend header
    const StaticKey* keys;     // GELF: sorted by key bytes
    uint32_t n_keys;
    // LTSV suffixes of bool, f64, i64, u64 (len 0xFFFFFFFF = not configured).  Named fields, not arrays: ONE dynamically indexed
    // member keeps the whole struct -- a kernel argument -- in scratch memory, and every read of any field then waits behind the
    // lane's output stores (the compiler also turns a chain of selects over array elements back into an indexed load)
    uint32_t suf_off0, suf_off1, suf_off2, suf_off3, suf_len0, suf_len1, suf_len2, suf_len3;
    FGE_HD void set_suffix(uint32_t k, uint32_t off, uint32_t len) {
        if (k == 0u) { suf_off0 = off; suf_len0 = len; }
        else if (k == 1u) { suf_off1 = off; suf_len1 = len; }
        else if (k == 2u) { suf_off2 = off; suf_len2 = len; }
        else { suf_off3 = off; suf_len3 = len; }
    }
    uint32_t src_fmt;          // which decoder produced the tables (fg_format)
    uint32_t enc;              // fg_encoder
    uint32_t merger;           // fg_merger
    uint32_t ltsv_extra_off, ltsv_extra_len;  // "k1:v1\tk2:v2" (already escaped, '_' stripped); len 0 = none
    uint32_t prepend_off, prepend_len;        // syslog_prepend_timestamp header; len 0xFFFFFFFF = not configured
    double now_ts;             // Record.ts of rows flagged FG_F_TS_NOW
    uint32_t sort_slots;       // GELF: entries of the per-lane ranking scratch (<= kSortSlots)
    uint64_t out_cap;          // write pass: nonzero = leave the output alone when out_offsets[n] exceeds it (fg_encode_device_async)
};

// encode status per line (fg_encode_error_string)
enum : uint32_t {
    ES_OK = 0,
    ES_DECODE_FAILED = 1,       // the row's decode status is an error: nothing to encode
    ES_5424_DATE = 2,           // "Failed to parse date"                       rfc5424_encoder.rs:46
    ES_5424_FORMAT = 3,         // "Failed to parse date as Rfc3339 format"     rfc5424_encoder.rs:52
    ES_3164_TS = 4,             // "Failed to parse unix timestamp in RFC3164 encoder"  rfc3164_encoder.rs:53
    ES_PASSTHROUGH_EMPTY = 5    // "Cannot output empty raw message"            passthrough_encoder.rs:47
};

namespace emit {

enum : uint32_t { M_RAW = 0, M_SD = 1, M_JSON = 2, M_JSON_RETRY = 3, M_WSJOIN = 4 };

FGE_HD uint32_t hexv(uint32_t c) {
    if (c - '0' <= 9u) return c - '0';
    c |= 0x20u;
    if (c - 'a' <= 5u) return c - 'a' + 10u;
    return 0;
}

// Streams the decoded bytes of rd[off .. off+len) into f(byte).
// Code size matters here: the emitters instantiate this once per call site with the sink's escaping inlined into f,
// so every mode produces "up to four decoded bytes of this step" (w, little-endian; nb) and ALL modes share ONE
// call site of f (M_RAW: its own).
template <class R, class F>
FGE_HD void for_each_decoded(R& rd, uint32_t off, uint32_t len, uint32_t mode, F&& f) {
    if (mode == M_RAW) {
        for (uint32_t i = 0; i < len; ++i) f(rd.byte(off + i));
        return;
    }
    const bool retry = mode == M_JSON_RETRY;
    bool esc = false;                  // M_SD: the previous byte was an unconsumed backslash
    bool in_tok = false, any = false;  // M_WSJOIN
    for (uint32_t i = 0; i < len;) {
        const uint32_t c = rd.byte(off + i);
        uint32_t w = c, nb = 1u;
        if (mode == M_SD) {  // unescape_sd_value: \" \\ \] lose the backslash, any other \x keeps both
            ++i;
            if (!esc) {
                if (c == '\\') {
                    esc = true;
                    nb = 0u;
                }
            } else {
                if (c != '"' && c != '\\' && c != ']') {
                    w = (uint32_t)'\\' | c << 8;
                    nb = 2u;
                }
                esc = false;
            }
        } else if (mode == M_WSJOIN) {  // str::split_whitespace(..).join(" ") (RFC3164 msg, rfc3164_decoder.rs:70)
            const uint32_t ws = r3164::ws_at(rd, off + i, off + len);
            if (ws) {
                in_tok = false;
                i += ws;
                nb = 0u;
            } else {
                ++i;
                if (!in_tok) {
                    if (any) {
                        w = (uint32_t)' ' | c << 8;
                        nb = 2u;
                    }
                    in_tok = any = true;
                }
            }
        } else if (c != '\\' || i + 1 >= len) {  // JSON escapes of an already validated string body: a plain byte
            ++i;
        } else {
            const uint32_t e = rd.byte(off + i + 1);
            i += 2;
            if (retry && e == '\n') {  // the reference replaced LF by "\\n": an escaped backslash, then 'n'
                w = (uint32_t)'\\' | (uint32_t)'n' << 8;
                nb = 2u;
            } else if (e == 'b') w = 8u;
            else if (e == 'f') w = 12u;
            else if (e == 'n') w = 10u;
            else if (e == 'r') w = 13u;
            else if (e == 't') w = 9u;
            else if (e == 'u') {
                if (i + 4 > len) {
                    nb = 0u;
                } else {
                    uint32_t n1 = hexv(rd.byte(off + i)) << 12 | hexv(rd.byte(off + i + 1)) << 8 | hexv(rd.byte(off + i + 2)) << 4 | hexv(rd.byte(off + i + 3));
                    i += 4;
                    if (n1 >= 0xD800u && n1 <= 0xDBFFu && i + 6 <= len) {
                        const uint32_t n2 = hexv(rd.byte(off + i + 2)) << 12 | hexv(rd.byte(off + i + 3)) << 8 | hexv(rd.byte(off + i + 4)) << 4 | hexv(rd.byte(off + i + 5));
                        i += 6;
                        n1 = (((n1 - 0xD800u) << 10) | (n2 - 0xDC00u)) + 0x10000u;
                    }
                    if (n1 < 0x80u) {
                        w = n1;
                    } else if (n1 < 0x800u) {
                        w = (0xC0u | (n1 >> 6)) | (0x80u | (n1 & 0x3Fu)) << 8;
                        nb = 2u;
                    } else if (n1 < 0x10000u) {
                        w = (0xE0u | (n1 >> 12)) | (0x80u | ((n1 >> 6) & 0x3Fu)) << 8 | (0x80u | (n1 & 0x3Fu)) << 16;
                        nb = 3u;
                    } else {
                        w = (0xF0u | (n1 >> 18)) | (0x80u | ((n1 >> 12) & 0x3Fu)) << 8 | (0x80u | ((n1 >> 6) & 0x3Fu)) << 16 | (0x80u | (n1 & 0x3Fu)) << 24;
                        nb = 4u;
                    }
                }
            } else {
                w = e;  // " \ /
            }
        }
        for (uint32_t j = 0; j < nb; ++j) f((w >> (8u * j)) & 0xFFu);
    }
}

// ---- sinks.  Protocol: put(c) one byte; put_word(w, nb) nb = 1..4 bytes, little-endian in w, the bytes above nb zero;
//      finish().  kCount sinks only count (add(n) = n bytes whose values do not matter). -----------------------------
struct CountSink {
    static constexpr bool kCount = true;
    uint32_t n = 0;
    uint32_t slow = 0;  // spans that held a byte to escape (Base::copy_raw): 0 = the write pass may copy this row's spans untested
    FGE_HD void mark_slow() { slow = 1u; }
    FGE_HD void put(uint32_t) { ++n; }
    FGE_HD void put_word(uint32_t, uint32_t nb) { n += nb; }
    FGE_HD void put16(uint32_t, uint32_t, uint32_t, uint32_t) { n += 16u; }
    FGE_HD void put_part(uint32_t, uint32_t, uint32_t, uint32_t, uint32_t nb) { n += nb; }
    FGE_HD void add(uint32_t k) { n += k; }
    FGE_HD void finish() {}
};
// Packs the byte stream into ALIGNED 16-BYTE stores (round 5; rounds 1-4: aligned dword stores).  A lane streams its own message, so
// one store instruction of the wave touches 64 different cache lines whatever its width -- the store path handles those one line per
// cycle: per-lane dword stores ran the write kernel at 0.9 TB/s of output (2.6 ms for the 2.4 GB of the 4 M-line corpus, the whole
// difference between the count pass and the write pass), per-lane 16-byte stores move the same bytes in a quarter of the store
// instructions: 3.0 TB/s (tools/probe/store_patterns.cpp, profiles/r05c_store_patterns.log; staging the bytes through LDS for
// wave-cooperative flushes measured SLOWER than that: 2.5 TB/s).  Messages of neighbouring lines are adjacent in the output and are
// written by other lanes at the same time, so nothing outside [q, q + length) may be touched: the first and the last 16-byte block of
// a message are stored by dwords and bytes, only over bytes that are the message's own.
struct PackSink {
    static constexpr bool kCount = false;
    struct alignas(16) Block { uint32_t x, y, z, w; };
    uint8_t* p;     // 16-byte aligned address of the block being assembled (after finish(): the end of the message)
    uint32_t b0 = 0, b1 = 0, b2 = 0, b3 = 0;  // its completed dwords
    uint32_t acc = 0;  // the dword being assembled: its low (k & 3) bytes are valid (placeholders where `head` says so)
    uint32_t k;        // bytes of the block that are filled, 0..15
    uint32_t head;     // bytes at the start of the FIRST block that belong to the previous message (0 once that block is out)
    FGE_HD explicit PackSink(uint8_t* q) {
        head = (uint32_t)((uintptr_t)q & 15u);
        p = q - head;
        k = head;
    }
    FGE_HD uint32_t dword_of(uint32_t d) const { return d == 0u ? b0 : d == 1u ? b1 : d == 2u ? b2 : b3; }
    // (the output lies in global memory: said explicitly, or the kernels' stores are flat_ instructions -- a slower address path that
    //  also counts against the LDS wait counter)
    template <class T>
    static FGE_HD void st(uint8_t* q, const T& v) {
#if defined(__HIP_DEVICE_COMPILE__)
        *reinterpret_cast<T __attribute__((address_space(1)))*>((uint8_t __attribute__((address_space(1)))*)q) = v;
#else
        *reinterpret_cast<T*>(q) = v;
#endif
    }
    // bytes [lo, hi) of the block at q, from its completed dwords (index < full) and `ac` (index == full): whole dwords as dwords.
    // A message's first and last block only -- ONE copy of the code per kernel, by value (inlined at each of the emitters' ~150 sink
    // calls it was most of the write kernel's 74 000 vector instructions).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(FG_EMIT_INLINE_STORE_PART)  // (the macro: an A/B build)
    static __device__ __attribute__((noinline))
#elif defined(__HIP_DEVICE_COMPILE__)
    static __device__ __forceinline__
#else
    static inline
#endif
    void store_part_at(uint8_t* q, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t ac, uint32_t lo, uint32_t hi, uint32_t full) {
#ifdef __HIP_DEVICE_COMPILE__
#pragma unroll
#endif
        for (uint32_t d = 0; d < 4u; ++d) {
            const uint32_t a = d * 4u > lo ? d * 4u : lo, e = d * 4u + 4u < hi ? d * 4u + 4u : hi;
            if (a >= e) continue;
            const uint32_t v = d < full ? (d == 0u ? w0 : d == 1u ? w1 : d == 2u ? w2 : w3) : ac;
            if (e - a == 4u) {
                st<uint32_t>(q + d * 4u, v);
            } else {
                for (uint32_t i = a; i < e; ++i) st<uint8_t>(q + i, (uint8_t)(v >> (8u * (i & 3u))));
            }
        }
    }
    FGE_HD void store_part(uint32_t lo, uint32_t hi, uint32_t full) { store_part_at(p, b0, b1, b2, b3, acc, lo, hi, full); }
    static FGE_HD void st16(uint8_t* q, uint32_t x, uint32_t y, uint32_t z, uint32_t w) {
#if defined(__HIP_DEVICE_COMPILE__)
        typedef uint32_t v4u __attribute__((ext_vector_type(4)));  // (a vector, not a struct: an aggregate copy forgets the address space)
        const v4u v = {x, y, z, w};
        *reinterpret_cast<v4u __attribute__((address_space(1)))*>((uint8_t __attribute__((address_space(1)))*)q) = v;
#else
        *reinterpret_cast<Block*>(q) = Block{x, y, z, w};
#endif
    }
    // 32-bit arithmetic only (64-bit shifts are slow on the vector ALU): w's low bytes complete the dword, its high
    // bytes start the next one
    FGE_HD void put_word(uint32_t w, uint32_t nb) {
        const uint32_t kb = k & 3u, sh = 8u * kb;
        const uint32_t lo = acc | (w << sh);
        if (kb + nb >= 4u) {
            const uint32_t d = k >> 2;  // (selects, not an array: a dynamically indexed local array lives in scratch memory)
            b0 = d == 0u ? lo : b0;
            b1 = d == 1u ? lo : b1;
            b2 = d == 2u ? lo : b2;
            b3 = d == 3u ? lo : b3;
            acc = sh ? w >> (32u - sh) : 0u;
            k += nb;
            if (k >= 16u) {  // the block is complete
                if (head) {
                    store_part(head, 16u, 4u);
                    head = 0;
                } else {
                    st16(p, b0, b1, b2, b3);
                }
                p += 16;
                k -= 16u;
            }
        } else {
            acc = lo;
            k += nb;
        }
    }
    // bytes [4 - sh, 8 - sh) of the eight bytes lo | hi << 32: the dword that a left shift by sh bytes moves across a dword boundary
    static FGE_HD uint32_t carry_bytes(uint32_t hi, uint32_t lo, uint32_t sh) {
#if defined(__HIP_DEVICE_COMPILE__)
        return __builtin_amdgcn_perm(hi, lo, 0x03020100u + (4u - sh) * 0x01010101u);  // (one v_perm_b32; 64-bit shifts are slow)
#else
        return (uint32_t)((((uint64_t)hi << 32) | lo) >> (8u * (4u - sh)));
#endif
    }
    // SIXTEEN bytes at once (the span copies: one LDS round trip of input): the block position k does not change, so exactly ONE
    // block completes per call -- whatever k is -- and every lane of a wave that copies a span stores its 16 bytes in the SAME store
    // instruction.  (Four put_word calls complete the block at a lane-specific call: four store instructions with a quarter of the
    // lanes each, and the write kernel gained nothing from 16-byte stores that way -- profiles/r05d_*cfg1*.)
    FGE_HD void put16(uint32_t q0, uint32_t q1, uint32_t q2, uint32_t q3) {
        const uint32_t d = k >> 2, sh = k & 3u;
        const uint32_t t0 = carry_bytes(q0, 0u, sh), t1 = carry_bytes(q1, q0, sh), t2 = carry_bytes(q2, q1, sh), t3 = carry_bytes(q3, q2, sh),
                       t4 = carry_bytes(0u, q3, sh);
        const uint32_t a0 = acc | t0;  // (acc holds the low sh bytes, t0 the rest: they do not overlap)
        // the block: its first d dwords are there already, then a0, then t1 ...; what does not fit starts the next block
        const uint32_t o0 = d == 0u ? a0 : b0;
        const uint32_t o1 = d == 0u ? t1 : d == 1u ? a0 : b1;
        const uint32_t o2 = d == 0u ? t2 : d == 1u ? t1 : d == 2u ? a0 : b2;
        const uint32_t o3 = d == 0u ? t3 : d == 1u ? t2 : d == 2u ? t1 : a0;
        if (head) {
            b0 = o0;
            b1 = o1;
            b2 = o2;
            b3 = o3;
            store_part(head, 16u, 4u);
            head = 0;
        } else {
            st16(p, o0, o1, o2, o3);
        }
        p += 16;
        b0 = d == 1u ? t3 : d == 2u ? t2 : t1;
        b1 = d == 2u ? t3 : t2;
        b2 = t3;
        acc = t4;
    }
    // ONE TO SIXTEEN bytes at once, nb of them, in q0 .. q3 (little-endian; the bytes beyond nb must be ZERO): the tail of a span, a
    // key's text, a number's digits.  Branch-free but for the store: a message is ~85 short pieces (puts of one to four bytes) around
    // its two dozen 16-byte copies, the lanes of a wave hold their blocks at 64 different fill levels, so some lane completes a block
    // at EVERY short put and the wave runs both sides of put_word's branches every time -- fewer, larger pieces are what the write
    // pass's instruction count is made of.
    FGE_HD void put_part(uint32_t q0, uint32_t q1, uint32_t q2, uint32_t q3, uint32_t nb) {
        const uint32_t d = k >> 2, sh = k & 3u;
        const uint32_t t0 = carry_bytes(q0, 0u, sh), t1 = carry_bytes(q1, q0, sh), t2 = carry_bytes(q2, q1, sh), t3 = carry_bytes(q3, q2, sh),
                       t4 = carry_bytes(0u, q3, sh);
        const uint32_t a0 = acc | t0;
        // the stream's dwords 0 .. 7 from the block's start: the d complete ones, a0, t1 .. t4, zeros
        const uint32_t o0 = d == 0u ? a0 : b0;
        const uint32_t o1 = d == 0u ? t1 : d == 1u ? a0 : b1;
        const uint32_t o2 = d == 0u ? t2 : d == 1u ? t1 : d == 2u ? a0 : b2;
        const uint32_t o3 = d == 0u ? t3 : d == 1u ? t2 : d == 2u ? t1 : a0;
        const uint32_t n0 = d == 0u ? t4 : d == 1u ? t3 : d == 2u ? t2 : t1;
        const uint32_t n1 = d == 1u ? t4 : d == 2u ? t3 : d == 3u ? t2 : 0u;
        const uint32_t n2 = d == 2u ? t4 : d == 3u ? t3 : 0u;
        const uint32_t n3 = d == 3u ? t4 : 0u;
        const uint32_t k2 = k + nb;
        const bool full = k2 >= 16u;
        if (full) {
            if (head) {
                b0 = o0;
                b1 = o1;
                b2 = o2;
                b3 = o3;
                store_part(head, 16u, 4u);
                head = 0;
            } else {
                st16(p, o0, o1, o2, o3);
            }
            p += 16;
        }
        b0 = full ? n0 : o0;
        b1 = full ? n1 : o1;
        b2 = full ? n2 : o2;
        b3 = full ? n3 : o3;
        k = full ? k2 - 16u : k2;
        const uint32_t dn = k >> 2;  // (the dword being assembled: its bytes beyond k are zero because q's beyond nb are)
        acc = dn == 0u ? b0 : dn == 1u ? b1 : dn == 2u ? b2 : b3;
    }
    FGE_HD void put(uint32_t c) { put_word(c & 0xFFu, 1u); }
    FGE_HD void add(uint32_t) {}
    FGE_HD void mark_slow() {}
    FGE_HD void finish() {
        if (k > head) store_part(head, k, k >> 2);
        p += k;
        head = 0;
        k = 0;
        acc = 0;
    }
};

// SWAR byte tests on a dword (exact as booleans: nonzero iff some byte matches)
FGE_HD uint32_t swar_zero(uint32_t v) { return (v - 0x01010101u) & ~v & 0x80808080u; }
FGE_HD uint32_t swar_lt20(uint32_t w) { return (w - 0x20202020u) & ~w & 0x80808080u; }
FGE_HD uint32_t swar_has(uint32_t w, uint32_t c) { return swar_zero(w ^ (c * 0x01010101u)); }
enum : uint32_t { ESC_NONE = 0, ESC_JSON = 1, ESC_LTSV_VAL = 2 };
template <uint32_t ESC>
FGE_HD bool word_needs_bytes(uint32_t w) {
    if (ESC == ESC_JSON) return (swar_lt20(w) | swar_has(w, '"') | swar_has(w, '\\')) != 0;
    if (ESC == ESC_LTSV_VAL) return (swar_has(w, '\t') | swar_has(w, '\n')) != 0;
    return false;
}
// the same tests without their final mask (bit 7 of a byte set = candidate): the terms of several dwords are OR-ed and masked ONCE --
// one compare and one branch per sixteen bytes instead of four short-circuit tests
FGE_HD uint32_t swar_zero_raw(uint32_t v) { return (v - 0x01010101u) & ~v; }
template <uint32_t ESC>
FGE_HD uint32_t word_needs_raw(uint32_t w) {
    if (ESC == ESC_JSON) return ((w - 0x20202020u) & ~w) | swar_zero_raw(w ^ 0x22222222u) | swar_zero_raw(w ^ 0x5C5C5C5Cu);
    if (ESC == ESC_LTSV_VAL) return swar_zero_raw(w ^ 0x09090909u) | swar_zero_raw(w ^ 0x0A0A0A0Au);
    return 0u;
}
template <uint32_t ESC>
FGE_HD bool any16_needs_bytes(uint32_t q0, uint32_t q1, uint32_t q2, uint32_t q3) {
    return ((word_needs_raw<ESC>(q0) | word_needs_raw<ESC>(q1) | word_needs_raw<ESC>(q2) | word_needs_raw<ESC>(q3)) & 0x80808080u) != 0u;
}

// decimal text of v through f(char code), without a digit buffer (a dynamically indexed local array is scratch
// memory on the GPU): 2 + 9 + 9 digits, leading zeros suppressed
template <class F>
FGE_HD void u64_digits(uint64_t v, F&& f) {
    const uint32_t c0 = (uint32_t)(v % 1000000000ull);
    const uint64_t r = v / 1000000000ull;
    const uint32_t c1 = (uint32_t)(r % 1000000000ull), c2 = (uint32_t)(r / 1000000000ull);  // c2 <= 18
    bool started = false;
    for (uint32_t chunk = 0; chunk < 3u; ++chunk) {
        const uint32_t c = chunk == 0 ? c2 : chunk == 1 ? c1 : c0;
        for (uint32_t p = chunk == 0 ? 10u : 100000000u; p; p /= 10u) {
            const uint32_t d = c / p % 10u;
            if (d || started || (chunk == 2u && p == 1u)) {
                f((uint32_t)'0' + d);
                started = true;
            }
        }
    }
}

// A line's table row in registers.  The kernel fetches it BEFORE it stages the tile (every load in flight, hidden
// behind the staging) and hands it to the emitter; without it the emitter fetches the row itself (load_row()).
struct RowRegs {
    fg_span s0, s1, s2, s3, s4, s5;
    double ts;
    uint32_t ef, ec;
    uint32_t plain = 0;  // write pass: the count pass found no byte to escape in any span it tested (row_size's *plain)
    FGE_HD void load(const DevTables& t, uint64_t li) {
        s0 = t.span[0][li];
        s1 = t.span[1][li];
        s2 = t.span[2][li];
        s3 = t.span[3][li];
        s4 = t.span[4][li];
        s5 = t.span[5][li];
        ts = t.ts[li];
        ef = t.ent_first[li];
        ec = t.ent_count[li];
    }
};

// a sink over a per-byte functor (a pair's value on its way through an escaper): pieces are taken apart again
template <class F>
struct FnSinkT {
    static constexpr bool kCount = false;
    F& f;
    FGE_HD void put(uint32_t c) { f(c); }
    FGE_HD void add(uint32_t) {}
    FGE_HD void put_part(uint32_t q0, uint32_t q1, uint32_t q2, uint32_t q3, uint32_t nb) {
        for (uint32_t i = 0; i < nb; ++i) f(((i < 4u ? q0 : i < 8u ? q1 : i < 12u ? q2 : q3) >> (8u * (i & 3u))) & 0xFFu);
    }
    FGE_HD void put16(uint32_t q0, uint32_t q1, uint32_t q2, uint32_t q3) { put_part(q0, q1, q2, q3, 16u); }
};

// Everything the emitters share: the row, the field views, the number formats.
template <class S, class R>
struct Base {
    S& out;
    const EncCfg& cfg;
    R rd;  // the line's bytes
    const DevTables& t;
    uint64_t li;
    uint32_t meta;
    // the line's table row, fetched ONCE with every load in flight (load_row(), first thing the emitters do): fetched
    // where they are used, each of the ~10 fields costs the wave a dependent HBM round trip
    fg_span s0{}, s1{}, s2{}, s3{}, s4{}, s5{};
    double row_ts = 0.0;
    uint32_t row_ef = 0, row_ec = 0;
    // The count pass has tested every span copy_raw moves and found no byte that needs escaping: the write pass copies them untested
    // (the three SWAR tests are 52 of the ~95 instructions a 16-byte step costs).  False = test, as the count pass itself does.
    bool plain = false;
    FGE_HD void load_row(const RowRegs* pre) {
        if (pre) {
            plain = pre->plain != 0u;
            s0 = pre->s0;
            s1 = pre->s1;
            s2 = pre->s2;
            s3 = pre->s3;
            s4 = pre->s4;
            s5 = pre->s5;
            row_ts = pre->ts;
            row_ef = pre->ef;
            row_ec = pre->ec;
            return;
        }
        s0 = t.span[0][li];
        s1 = t.span[1][li];
        s2 = t.span[2][li];
        s3 = t.span[3][li];
        s4 = t.span[4][li];
        s5 = t.span[5][li];
        row_ts = t.ts[li];
        row_ef = t.ent_first[li];
        row_ec = t.ent_count[li];
    }
    FGE_HD fg_span span(int col) const {  // scalar selects (a ternary chain over the structs kept the emitter in memory)
        fg_span r;
        r.off = col == 0 ? s0.off : col == 1 ? s1.off : col == 2 ? s2.off : col == 3 ? s3.off : col == 4 ? s4.off : s5.off;
        r.len = col == 0 ? s0.len : col == 1 ? s1.len : col == 2 ? s2.len : col == 3 ? s3.len : col == 4 ? s4.len : s5.len;
        return r;
    }

    FGE_HD uint32_t flags() const { return FG_META_FLAGS(meta); }
    FGE_HD uint32_t json_mode() const { return (flags() & FG_F_GELF_RETRY) ? (uint32_t)M_JSON_RETRY : (uint32_t)M_JSON; }
    // decode mode of a top-level string field
    FGE_HD uint32_t field_mode(int col) const {
        if (cfg.src_fmt == FG_RFC3164) return (col == S_MSG && (flags() & FG_F_MSG_JOIN)) ? (uint32_t)M_WSJOIN : (uint32_t)M_RAW;
        if (cfg.src_fmt != FG_GELF) return M_RAW;
        const uint32_t bit = col == S_HOST ? FG_F_HOST_ESC : col == S_MSG ? FG_F_MSG_ESC : col == S_FULL ? FG_F_FULLMSG_ESC : 0u;
        return (flags() & bit) ? json_mode() : (uint32_t)M_RAW;
    }
    FGE_HD uint32_t value_mode(uint32_t e) const {
        if (!(t.ent_flags[e] & FG_EF_VAL_ESC)) return M_RAW;
        return cfg.src_fmt == FG_RFC5424 ? (uint32_t)M_SD : cfg.src_fmt == FG_GELF ? json_mode() : (uint32_t)M_RAW;
    }
    FGE_HD double record_ts() const { return (flags() & FG_F_TS_NOW) ? cfg.now_ts : row_ts; }

    // a literal (n <= 32 bytes, known at the call site: the words below fold into constants), as ONE piece per sixteen bytes -- and
    // with the byte `lead` in front of it when the caller has one (a separator: a put of its own costs what a piece does)
    FGE_HD void lit(const char* s, uint32_t n, uint32_t lead = 0u) {
        if (S::kCount) {
            out.add(n + (lead ? 1u : 0u));
            return;
        }
        uint32_t q[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u}, p[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};  // without / behind the lead byte
        for (uint32_t i = 0; i < n && i < 31u; ++i) {
            q[i >> 2] |= (uint32_t)(uint8_t)s[i] << (8u * (i & 3u));
            p[(i + 1u) >> 2] |= (uint32_t)(uint8_t)s[i] << (8u * ((i + 1u) & 3u));
        }
        const bool ld = lead != 0u;
        const uint32_t m = n + (ld ? 1u : 0u);
        const uint32_t w0 = ld ? (p[0] | lead) : q[0], w1 = ld ? p[1] : q[1], w2 = ld ? p[2] : q[2], w3 = ld ? p[3] : q[3];
        if (m <= 16u) {
            out.put_part(w0, w1, w2, w3, m);
        } else {
            out.put16(w0, w1, w2, w3);
            out.put_part(ld ? p[4] : q[4], ld ? p[5] : q[5], ld ? p[6] : q[6], ld ? p[7] : q[7], m - 16u);
        }
    }
    // a piece of the configuration blob; every piece starts 4-byte aligned and is followed by readable padding
    FGE_HD void blob(uint32_t off, uint32_t len) {
        if (S::kCount) {
            out.add(len);
            return;
        }
        const uint32_t* w = reinterpret_cast<const uint32_t*>(cfg.blob + off);
        uint32_t i = 0;
        for (; i + 16u <= len; i += 16u) out.put16(w[i >> 2], w[(i >> 2) + 1u], w[(i >> 2) + 2u], w[(i >> 2) + 3u]);
        if (i < len) {  // (the blob ends in sixteen bytes of padding: fg_enc_cfg.hpp)
            uint32_t q[4] = {w[i >> 2], w[(i >> 2) + 1u], w[(i >> 2) + 2u], w[(i >> 2) + 3u]};
            keep_low(q, len - i);
            out.put_part(q[0], q[1], q[2], q[3], len - i);
        }
    }
    // rd[off .. off+len) copied through the per-byte functor `fb` only where a dword holds a byte that needs it
    // (ESC): everything else moves four bytes at a time
    // the low nb (0 .. 16) of the sixteen bytes q[0 .. 3]; the rest zero
    static FGE_HD void keep_low(uint32_t* q, uint32_t nb) {
#ifdef __HIP_DEVICE_COMPILE__
#pragma unroll
#endif
        for (uint32_t j = 0; j < 4u; ++j) {
            const uint32_t v = nb > 4u * j ? nb - 4u * j : 0u;  // this dword's bytes
            q[j] = v >= 4u ? q[j] : (q[j] & ((1u << (8u * v)) - 1u));
        }
    }
    // byte c becomes byte `at` (0 .. 15) of the sixteen bytes q[0 .. 3] (which is zero there)
    static FGE_HD void set_byte(uint32_t* q, uint32_t at, uint32_t c) {
        const uint32_t w = c << (8u * (at & 3u)), j = at >> 2;
        q[0] |= j == 0u ? w : 0u;
        q[1] |= j == 1u ? w : 0u;
        q[2] |= j == 2u ? w : 0u;
        q[3] |= j == 3u ? w : 0u;
    }
    // rd[off .. off+len) copied through the per-byte functor `fb` only where a dword holds a byte that needs it
    // (ESC): everything else moves sixteen bytes at a time, the span's last 1 .. 15 bytes as ONE piece -- with the byte `sfx` behind
    // them when the caller has one (a string's closing quote: a put of its own costs what a whole piece does, PackSink::put_part)
    template <uint32_t ESC, class FB>
    FGE_HD void copy_raw(uint32_t off, uint32_t len, FB&& fb, uint32_t sfx = 0u) {
        if (ESC == ESC_NONE && S::kCount) {
            out.add(len + (sfx ? 1u : 0u));
            return;
        }
        uint32_t i = 0;
        for (; i + 16u <= len; i += 16u) {  // 16 bytes per LDS round trip (the copy is latency-bound: one wave or two per SIMD)
            uint32_t q[4];
            rd.load16(off + i, q);
            if (plain || !any16_needs_bytes<ESC>(q[0], q[1], q[2], q[3])) {
                out.put16(q[0], q[1], q[2], q[3]);
                continue;
            }
            out.mark_slow();
            slow16<ESC>(q, 16u, fb);
        }
        const uint32_t nb = len - i;  // 0 .. 15
        if (nb == 0u && !sfx) return;
        uint32_t q[4] = {0u, 0u, 0u, 0u};
        if (nb) {
            rd.load16p(off + i, nb, q);
            keep_low(q, nb);
            // (the bytes beyond nb are zero: for the escape test they read as plain letters)
            const uint32_t fill = 0x61616161u;
            const uint32_t f0 = nb >= 4u ? 0u : fill << (8u * nb), f1 = nb >= 8u ? 0u : nb <= 4u ? fill : fill << (8u * (nb - 4u)),
                           f2 = nb >= 12u ? 0u : nb <= 8u ? fill : fill << (8u * (nb - 8u)), f3 = nb <= 12u ? fill : fill << (8u * (nb - 12u));
            if (!plain && any16_needs_bytes<ESC>(q[0] | f0, q[1] | f1, q[2] | f2, q[3] | f3)) {
                out.mark_slow();
                slow16<ESC>(q, nb, fb);
                if (sfx) out.put(sfx);
                return;
            }
        }
        if (sfx) set_byte(q, nb, sfx);
        out.put_part(q[0], q[1], q[2], q[3], nb + (sfx ? 1u : 0u));
    }
    // nb (1 .. 16) bytes in q with a byte that needs fb among them: dword by dword, byte-wise only where it is
    template <uint32_t ESC, class FB>
    FGE_HD void slow16(const uint32_t* q, uint32_t nb, FB&& fb) {
        for (uint32_t j = 0; j < 4u && 4u * j < nb; ++j) {
            const uint32_t w = j == 0u ? q[0] : j == 1u ? q[1] : j == 2u ? q[2] : q[3];
            const uint32_t m = nb - 4u * j < 4u ? nb - 4u * j : 4u;
            if (m == 4u && !word_needs_bytes<ESC>(w)) {
                out.put_word(w, 4u);
                continue;
            }
            for (uint32_t b = 0; b < m; ++b) fb((w >> (8u * b)) & 0xFFu);
        }
    }
    FGE_HD void raw_field(int col, uint32_t sfx = 0u) {  // a top-level field, decoded, unmodified [+ the separator behind it]
        const fg_span s = this->span(col);
        const uint32_t mode = field_mode(col);
        if (mode == M_RAW) {
            copy_raw<ESC_NONE>(s.off, s.len, [&](uint32_t c) { out.put(c); }, sfx);
            return;
        }
        for_each_decoded(rd, s.off, s.len, mode, [&](uint32_t c) { out.put(c); });
        if (sfx) out.put(sfx);
    }
    // two decimal digits as two ASCII bytes (the tens in the low byte), v < 100
    static FGE_HD uint32_t d2(uint32_t v) { return ((uint32_t)'0' + v / 10u) | ((uint32_t)'0' + v % 10u) << 8; }
    FGE_HD void u64_text(uint64_t v) {
        if (v < 1000u) {  // level, facility, small typed values: the common case, one piece
            const uint32_t x = (uint32_t)v, nd = x >= 100u ? 3u : x >= 10u ? 2u : 1u;
            const uint32_t h = (uint32_t)'0' + x / 100u, m = (uint32_t)'0' + x / 10u % 10u, l = (uint32_t)'0' + x % 10u;
            out.put_part(nd == 3u ? (h | m << 8 | l << 16) : nd == 2u ? (m | l << 8) : l, 0u, 0u, 0u, nd);
            return;
        }
        u64_digits(v, [&](uint32_t c) { out.put(c); });
    }
    FGE_HD void i64_text(int64_t x) {
        if (x < 0) {
            out.put('-');
            u64_text(0ull - (uint64_t)x);
        } else {
            u64_text((uint64_t)x);
        }
    }
    FGE_HD void pad2(uint32_t v) {
        out.put('0' + v / 10u);
        out.put('0' + v % 10u);
    }
    // <pri>: ((facility << 3) & 0xF8) + (severity & 7) in u8 arithmetic
    FGE_HD void pri() {  // "<" + 1..3 digits + ">": one piece
        const uint32_t npri = (((FG_META_FACILITY(meta) << 3) & 0xF8u) + (FG_META_SEVERITY(meta) & 7u)) & 0xFFu;
        const uint32_t nd = npri >= 100u ? 3u : npri >= 10u ? 2u : 1u;
        const uint32_t h = npri / 100u, m = npri / 10u % 10u, l = npri % 10u;
        // bytes: '<' d.. '>'
        const uint32_t digits = nd == 3u ? (('0' + h) | ('0' + m) << 8 | ('0' + l) << 16) : nd == 2u ? (('0' + m) | ('0' + l) << 8) : ('0' + l);
        const uint32_t w0 = (uint32_t)'<' | digits << 8;                 // '<' + up to three digits
        const uint32_t gt = (uint32_t)'>';
        // '>' lands at byte 1 + nd: in w0 for nd <= 2, in w1 for nd == 3
        out.put_part(nd == 3u ? w0 : (w0 | gt << (8u * (1u + nd))), nd == 3u ? gt : 0u, 0u, 0u, nd + 2u);
    }
    FGE_HD bool has_pri() const { return FG_META_FACILITY(meta) != 0xFFu && FG_META_SEVERITY(meta) != 0xFFu; }

    // ---- entry names: the Record key is '_' + name [+ LTSV suffix]; for GELF-decoded records the '_' is only added
    //      when the (decoded) key does not start with one (gelf_decoder.rs:99-103) ------------------------------------
    struct Dyn {
        uint32_t off, len;  // raw name span
        uint32_t mode;      // M_RAW / M_JSON*
        uint32_t skip;      // decoded bytes to skip: 1 when a GELF key already starts with '_'
        uint32_t dlen;      // decoded length after the skip
        uint32_t so, sl;    // suffix in the blob (sl = 0: none)
    };
    FGE_HD Dyn dyn_of(uint32_t e) {
        const fg_span nm = t.ent_name[e];
        Dyn d{nm.off, nm.len, M_RAW, 0u, nm.len, 0u, 0u};
        const uint32_t ty = t.ent_type[e];
        const uint32_t ef = t.ent_flags[e];
        if (cfg.src_fmt == FG_GELF) {
            if (ef & FG_EF_NAME_ESC) {
                d.mode = json_mode();
                uint32_t n = 0, first = 0x100u;
                for_each_decoded(rd, nm.off, nm.len, d.mode, [&](uint32_t c) {
                    if (n == 0) first = c;
                    ++n;
                });
                d.skip = first == '_' ? 1u : 0u;
                d.dlen = n - d.skip;
            } else if (nm.len && rd.byte(nm.off) == '_') {
                d.skip = 1;
                d.dlen = nm.len - 1u;
            }
        }
        const uint32_t si = ty - FG_T_BOOL;
        // (mask-and-or, each term ONE field: `c ? field_a : field_b` is folded into a load through a selected pointer before
        //  this function is inlined into the kernel -- which is the dynamically indexed access again)
        const uint32_t m0 = 0u - (uint32_t)(si == 0u), m1 = 0u - (uint32_t)(si == 1u), m2 = 0u - (uint32_t)(si == 2u), m3 = 0u - (uint32_t)(si == 3u);
        const uint32_t sl = (cfg.suf_len0 & m0) | (cfg.suf_len1 & m1) | (cfg.suf_len2 & m2) | (cfg.suf_len3 & m3);
        const uint32_t so = (cfg.suf_off0 & m0) | (cfg.suf_off1 & m1) | (cfg.suf_off2 & m2) | (cfg.suf_off3 & m3);
        if ((ef & FG_EF_SUFFIX) && ty >= FG_T_BOOL && ty <= FG_T_U64 && sl != 0xFFFFFFFFu) {
            d.so = so;
            d.sl = sl;
        }
        return d;
    }
    // streams the key WITHOUT its leading '_' (name after the skip, then the suffix)
    template <class F>
    FGE_HD void dyn_stream(const Dyn& d, F&& f) {
        uint32_t k = 0;
        for_each_decoded(rd, d.off, d.len, d.mode, [&](uint32_t c) {
            if (k++ >= d.skip) f(c);
        });
        for (uint32_t i = 0; i < d.sl; ++i) f((uint32_t)cfg.blob[d.so + i]);
    }
    // byte k of the key WITHOUT its leading '_' (k < dlen + sl); escaped names (rare) decode from the start
    FGE_HD uint32_t dyn_byte(const Dyn& d, uint32_t k) {
        if (k >= d.dlen) return cfg.blob[d.so + (k - d.dlen)];
        if (d.mode == M_RAW) return rd.byte(d.off + d.skip + k);
        uint32_t i = 0, r = 0;
        const uint32_t want = k + d.skip;
        for_each_decoded(rd, d.off, d.len, d.mode, [&](uint32_t c) {
            if (i++ == want) r = c;
        });
        return r;
    }
    // a pair's value as `{}` prints it (impl Display for SDValue's payload; Null prints nothing)
    template <class F>
    FGE_HD void value_display(uint32_t e, F&& f) {
        const uint32_t ty = t.ent_type[e];
        const uint64_t v = t.ent_val[e];
        FnSinkT<F> fs{f};
        if (ty == FG_T_STRING) {
            for_each_decoded(rd, (uint32_t)v, (uint32_t)(v >> 32), value_mode(e), f);
        } else if (ty == FG_T_BOOL) {
            const char* s = v ? "true" : "false";
            for (uint32_t i = 0; s[i]; ++i) f((uint32_t)(uint8_t)s[i]);
        } else if (ty == FG_T_F64) {
            double d;
            memcpy(&d, &v, 8);
            shortest::display_f64(d, fs);
        } else if (ty == FG_T_I64 || ty == FG_T_U64) {
            uint64_t m = v;
            if (ty == FG_T_I64 && (int64_t)v < 0) {
                f((uint32_t)'-');
                m = 0ull - v;
            }
            u64_digits(m, f);
        }
    }
    // every StructuredData of the record through `impl Display` (record.rs:42-67), concatenated
    FGE_HD void sd_display() {
        const uint32_t first = this->row_ef, cnt = this->row_ec;
        bool open = false;
        for (uint32_t e = first; e < first + cnt; ++e) {
            if (t.ent_type[e] == FG_T_SDID) {
                if (open) out.put(']');
                out.put('[');
                open = true;
                const fg_span id = t.ent_name[e];
                for (uint32_t i = 0; i < id.len; ++i) out.put(rd.byte(id.off + i));
                continue;
            }
            if (!open) {  // LTSV / GELF records: one element, sd_id None
                out.put('[');
                open = true;
            }
            out.put(' ');
            const Dyn d = dyn_of(e);
            dyn_stream(d, [&](uint32_t c) { out.put(c); });
            if (t.ent_type[e] != FG_T_NULL) {
                out.put('=');
                out.put('"');
                if (t.ent_type[e] == FG_T_STRING && value_mode(e) == M_RAW) {
                    const uint64_t v = t.ent_val[e];
                    copy_raw<ESC_NONE>((uint32_t)v, (uint32_t)(v >> 32), [&](uint32_t c) { out.put(c); });
                } else {
                    value_display(e, [&](uint32_t c) { out.put(c); });
                }
                out.put('"');
            }
        }
        if (open) out.put(']');
    }
    FGE_HD bool has_sd() const { return this->row_ec != 0; }
    FGE_HD bool some(int col) const { return this->span(col).len != FG_NONE; }
};

// =================================================================================================
// GELF
// =================================================================================================
constexpr uint32_t kSortSlots = 32;

template <class S, class R>
struct GelfEmitter : Base<S, R> {
    using B = Base<S, R>;
    using typename B::Dyn;
    using B::cfg;
    using B::li;
    using B::meta;
    using B::out;
    using B::rd;
    using B::t;
    bool first_member = true;

    FGE_HD GelfEmitter(S& o, const EncCfg& c, R r, const DevTables& tb, uint64_t l, uint32_t m, const RowRegs* pre) : B{o, c, r, tb, l, m} { this->load_row(pre); }

    FGE_HD void esc_byte(uint32_t c) {  // serde_json 0.8 escape_str; one or two put_word sites (code size)
        uint32_t w = c, nb = 1u;
        if (c == '"' || c == '\\') {
            w = (uint32_t)'\\' | c << 8;
            nb = 2u;
        } else if (c < 0x20u) {
            const uint32_t x = c == 8u ? 'b' : c == 9u ? 't' : c == 10u ? 'n' : c == 12u ? 'f' : c == 13u ? 'r' : 0u;
            if (x) {
                w = (uint32_t)'\\' | x << 8;
            } else {  // \u00XX
                out.put_word((uint32_t)'\\' | (uint32_t)'u' << 8 | (uint32_t)'0' << 16 | (uint32_t)'0' << 24, 4u);
                const uint32_t lo = c & 15u;
                w = (c >> 4 ? (uint32_t)'1' : (uint32_t)'0') | (lo < 10u ? '0' + lo : 'a' + lo - 10u) << 8;
            }
            nb = 2u;
        }
        out.put_word(w, nb);
    }
    FGE_HD void member_start() {  // (the object's '{' comes with its first member)
        out.put(first_member ? (uint32_t)'{' : (uint32_t)',');
        first_member = false;
    }
    FGE_HD void key_static(const StaticKey& k) {  // `,"key":` / `{"key":` (StaticKey: with everything behind it that is constant)
        this->blob(first_member ? k.text1_off : k.text_off, k.text_len);
        first_member = false;
    }
    FGE_HD void str_span(uint32_t off, uint32_t len, uint32_t mode, bool open_quote = true) {
        if (open_quote) out.put('"');
        if (mode == M_RAW) {  // (the closing quote rides on the span's last piece)
            this->template copy_raw<ESC_JSON>(off, len, [&](uint32_t c) { esc_byte(c); }, (uint32_t)'"');
            return;
        }
        for_each_decoded(rd, off, len, mode, [&](uint32_t c) { esc_byte(c); });
        out.put('"');
    }
    FGE_HD void str_field(int col) {  // (a static key's value: the key's text ends in the opening quote)
        const fg_span s = this->span(col);
        str_span(s.off, s.len, this->field_mode(col), false);
    }
    FGE_HD void f64_text(double d) {
        uint64_t b;
        memcpy(&b, &d, 8);
        if (((b >> 52) & 0x7FFu) == 0x7FFu) {  // NaN / inf
            this->lit("null", 4);
            return;
        }
        dtoa::write_pieces(d, out);  // digits in registers, as one piece or two where the shape allows (no char buffer: that would be scratch memory)
    }
    FGE_HD int cmp_dyn(const Dyn& a, const Dyn& b) {
        const uint32_t la = a.dlen + a.sl, lb = b.dlen + b.sl, n = la < lb ? la : lb;
        for (uint32_t k = 0; k < n; ++k) {
            const uint32_t x = this->dyn_byte(a, k), y = this->dyn_byte(b, k);
            if (x != y) return x < y ? -1 : 1;
        }
        return la == lb ? 0 : (la < lb ? -1 : 1);
    }
    // full key ('_' + ...) against a static key
    FGE_HD int cmp_dyn_static(const Dyn& a, const StaticKey& s) {
        if (s.key_len == 0) return 1;
        const uint32_t s0 = cfg.blob[s.key_off];
        if (s0 != '_') return '_' < s0 ? -1 : 1;
        const uint32_t la = a.dlen + a.sl, lb = s.key_len - 1u, n = la < lb ? la : lb;
        for (uint32_t k = 0; k < n; ++k) {
            const uint32_t x = this->dyn_byte(a, k), y = cfg.blob[s.key_off + 1u + k];
            if (x != y) return x < y ? -1 : 1;
        }
        return la == lb ? 0 : (la < lb ? -1 : 1);
    }
    FGE_HD void emit_dyn(uint32_t e) {
        const Dyn d = this->dyn_of(e);
        member_start();
        out.put('"');
        out.put('_');
        this->dyn_stream(d, [&](uint32_t c) { esc_byte(c); });
        out.put('"');
        out.put(':');
        const uint32_t ty = t.ent_type[e];
        const uint64_t v = t.ent_val[e];
        if (ty == FG_T_STRING) {
            str_span((uint32_t)v, (uint32_t)(v >> 32), this->value_mode(e));
        } else if (ty == FG_T_BOOL) {
            if (v) this->lit("true", 4);
            else this->lit("false", 5);
        } else if (ty == FG_T_NULL) {
            this->lit("null", 4);
        } else if (ty == FG_T_U64) {
            this->u64_text(v);
        } else if (ty == FG_T_I64) {
            this->i64_text((int64_t)v);
        } else {
            double d2;
            memcpy(&d2, &v, 8);
            f64_text(d2);
        }
    }
    FGE_HD void emit_static(const StaticKey& k, uint32_t sdid_entry) {
        switch (k.kind) {
            case SK_APP: if (!this->some(S_APP)) return; break;
            case SK_FULL: if (!this->some(S_FULL)) return; break;
            case SK_PROC: if (!this->some(S_PROC)) return; break;
            case SK_LEVEL: if (FG_META_SEVERITY(meta) == 0xFFu) return; break;
            case SK_SDID: if (sdid_entry == 0xFFFFFFFFu) return; break;
            default: break;
        }
        key_static(k);
        int col = -1;  // a string field: ONE str_field call site below
        switch (k.kind) {
            case SK_APP: col = S_APP; break;
            case SK_FULL: col = S_FULL; break;
            case SK_PROC: col = S_PROC; break;
            case SK_HOST: {
                const fg_span s = this->span(S_HOST);
                if (s.len == 0u || s.len == FG_NONE) {
                    this->lit("unknown\"", 8);
                } else {
                    col = S_HOST;
                }
                break;
            }
            case SK_LEVEL: out.put('0' + FG_META_SEVERITY(meta)); break;
            case SK_SDID: {
                const fg_span id = t.ent_name[sdid_entry];
                for (uint32_t i = 0; i < id.len; ++i) esc_byte(rd.byte(id.off + i));
                out.put('"');
                break;
            }
            case SK_SHORT: {
                if (!this->some(S_MSG)) {
                    out.put('-');
                    out.put('"');
                } else {
                    col = S_MSG;
                }
                break;
            }
            case SK_TS: f64_text(this->record_ts()); break;
            default: break;  // SK_VERSION, SK_EXTRA: the value is part of the precomputed member text
        }
        if (col >= 0) str_field(col);
    }

    // keys64 / slot_ent / order: this lane's scratch (kSortSlots each)
    FGE_HD uint32_t run(uint64_t* keys64, uint8_t* slot_ent, uint8_t* order) {
        const uint32_t first = this->row_ef, cnt = this->row_ec;
        // pairs -> slots (and the LAST sd_id: every element's insert replaces the previous one)
        uint32_t sdid_entry = 0xFFFFFFFFu, np = 0;
        bool ranked = cnt <= 255u;
        for (uint32_t e = first; e < first + cnt; ++e) {
            if (t.ent_type[e] == FG_T_SDID) {
                sdid_entry = e;
                continue;
            }
            if (np < cfg.sort_slots && ranked) {
                const Dyn d = this->dyn_of(e);
                uint64_t pre = 0;
                for (uint32_t k = 0; k < 7u; ++k) pre = (pre << 8) | (k < d.dlen + d.sl ? this->dyn_byte(d, k) : 0u);
                keys64[np] = (pre << 8) | np;  // 7 key bytes big-endian, then the slot: equal keys keep insertion order
                slot_ent[np] = (uint8_t)(e - first);
            } else {
                ranked = false;
            }
            ++np;
        }
        if (ranked) {
            uint64_t k[kSortSlots];
#pragma unroll
            for (uint32_t j = 0; j < kSortSlots; ++j) k[j] = j < np ? keys64[j] : ~0ull;
            for (uint32_t i = 0; i < np; ++i) {  // (k[] stays in registers: only the inner loop is unrolled)
                const uint64_t ki = keys64[i];
                uint32_t rank = 0;
#pragma unroll
                for (uint32_t j = 0; j < kSortSlots; ++j) rank += k[j] < ki ? 1u : 0u;
                order[rank] = (uint8_t)i;
            }
            // adjacent equal 7-byte prefixes: duplicates (keep the later insert) or an unresolved order
            for (uint32_t r = 0; r + 1u < np && ranked; ++r) {
                const uint32_t sa = order[r], sb = order[r + 1u];
                if ((keys64[sa] >> 8) != (keys64[sb] >> 8)) continue;
                const int c = cmp_dyn(this->dyn_of(first + slot_ent[sa]), this->dyn_of(first + slot_ent[sb]));
                if (c == 0) order[r] = 0xFFu;
                else ranked = false;  // two different names share 7 bytes: exact selection below
            }
        }
        // ONE member loop for both orders (one emit_static / emit_dyn call site each: code size): the next entry in
        // key order comes from the ranking or, when that is not available, from an exact selection -- repeatedly the
        // smallest key greater than the previous one, among equal keys the LAST entry (the later insert)
        const uint32_t kNone = 0xFFFFFFFFu;
        uint32_t sk = 0;  // next static key
        uint32_t r = 0, prev = kNone;
        for (;;) {
            uint32_t e = kNone;
            if (ranked) {
                while (r < np && order[r] == 0xFFu) ++r;
                if (r < np) e = first + slot_ent[order[r++]];
            } else {
                for (uint32_t c = first; c < first + cnt; ++c) {
                    if (t.ent_type[c] == FG_T_SDID) continue;
                    const Dyn dc = this->dyn_of(c);
                    if (prev != kNone && cmp_dyn(dc, this->dyn_of(prev)) <= 0) continue;
                    if (e == kNone || cmp_dyn(dc, this->dyn_of(e)) <= 0) e = c;
                }
                prev = e;
            }
            // the static keys that sort before the entry (an equal one replaces it: gelf_extra is inserted last);
            // all the remaining ones once the entries are exhausted
            Dyn d{};
            if (e != kNone) d = this->dyn_of(e);
            bool shadowed = false;
            while (sk < cfg.n_keys) {
                if (e != kNone) {
                    const int c = cmp_dyn_static(d, cfg.keys[sk]);
                    if (c < 0) break;
                    if (c == 0) shadowed = true;
                }
                emit_static(cfg.keys[sk], sdid_entry);
                ++sk;
            }
            if (e == kNone) break;
            if (!shadowed) emit_dyn(e);
        }
        if (first_member) out.put('{');  // (no member at all)
        out.put('}');
        return ES_OK;
    }
};

// =================================================================================================
// LTSV
// =================================================================================================
template <class S, class R>
struct LtsvEmitter : Base<S, R> {
    using B = Base<S, R>;
    using B::cfg;
    using B::li;
    using B::meta;
    using B::out;
    using B::rd;
    using B::t;
    bool first = true;
    FGE_HD LtsvEmitter(S& o, const EncCfg& c, R r, const DevTables& tb, uint64_t l, uint32_t m, const RowRegs* pre) : B{o, c, r, tb, l, m} { this->load_row(pre); }

    FGE_HD void key_byte(uint32_t c) { out.put(c == '\n' || c == '\t' ? (uint32_t)' ' : c == ':' ? (uint32_t)'_' : c); }
    FGE_HD void val_byte(uint32_t c) { out.put(c == '\n' || c == '\t' ? (uint32_t)' ' : c); }
    // LTSVString::insert up to and including ':' -- `key` includes its ':' -- behind the TAB of every entry but the first: one piece
    FGE_HD void start(const char* key, uint32_t n) {
        this->lit(key, n, first ? 0u : (uint32_t)'\t');
        first = false;
    }
    FGE_HD void val_span(uint32_t off, uint32_t len, uint32_t mode) {
        if (mode == M_RAW) this->template copy_raw<ESC_LTSV_VAL>(off, len, [&](uint32_t c) { val_byte(c); });
        else for_each_decoded(rd, off, len, mode, [&](uint32_t c) { val_byte(c); });
    }
    FGE_HD void field(const char* key, uint32_t n, int col) {
        start(key, n);
        const fg_span s = this->span(col);
        val_span(s.off, s.len, this->field_mode(col));
    }
    FGE_HD uint32_t run() {
        const uint32_t ef = this->row_ef, cnt = this->row_ec;
        for (uint32_t e = ef; e < ef + cnt; ++e) {
            if (t.ent_type[e] == FG_T_SDID) continue;
            if (!first) out.put('\t');
            first = false;
            const typename B::Dyn d = this->dyn_of(e);
            this->dyn_stream(d, [&](uint32_t c) { key_byte(c); });
            out.put(':');
            if (t.ent_type[e] == FG_T_STRING) {
                const uint64_t v = t.ent_val[e];
                val_span((uint32_t)v, (uint32_t)(v >> 32), this->value_mode(e));
            } else {
                this->value_display(e, [&](uint32_t c) { val_byte(c); });
            }
        }
        if (cfg.ltsv_extra_len) {
            if (!first) out.put('\t');
            first = false;
            this->blob(cfg.ltsv_extra_off, cfg.ltsv_extra_len);
        }
        start("host:", 5);
        if (this->some(S_HOST)) {
            const fg_span s = this->span(S_HOST);
            val_span(s.off, s.len, this->field_mode(S_HOST));
        }
        start("time:", 5);
        shortest::display_f64(this->record_ts(), out);
        if (this->some(S_MSG)) field("message:", 8, S_MSG);
        if (this->some(S_FULL)) field("full_message:", 13, S_FULL);
        if (FG_META_SEVERITY(meta) != 0xFFu) {
            start("level:", 6);
            this->u64_text(FG_META_SEVERITY(meta));
        }
        if (FG_META_FACILITY(meta) != 0xFFu) {
            start("facility:", 9);
            this->u64_text(FG_META_FACILITY(meta));
        }
        if (this->some(S_APP)) field("appname:", 8, S_APP);
        if (this->some(S_PROC)) field("procid:", 7, S_PROC);
        if (this->some(S_MSGID)) field("msgid:", 6, S_MSGID);
        return ES_OK;
    }
};

// =================================================================================================
// RFC5424 / RFC3164 / passthrough
// =================================================================================================
constexpr int64_t kMinUnix = -377705116800ll, kMaxUnix = 253402300799ll;  // time 0.3: years -9999 ..= 9999

// ((ts * 1000.0) as i128) * 1_000_000 (release build: saturating cast, wrapping multiplication), then
// OffsetDateTime::from_unix_timestamp_nanos: false = out of range ("Failed to parse date").
FGE_HD bool rfc5424_ts_split(double ts, int64_t* secs, uint32_t* nanos) {
    const double x = ts * 1000.0;
    // i128 two's complement as (hi, lo)
    uint64_t hi, lo;
    if (x != x) {
        hi = lo = 0;
    } else if (x >= 170141183460469231731687303715884105728.0) {
        hi = 0x7FFFFFFFFFFFFFFFull;
        lo = ~0ull;
    } else if (x <= -170141183460469231731687303715884105728.0) {
        hi = 0x8000000000000000ull;
        lo = 0;
    } else {
        const double a = x < 0 ? -x : x;
        uint64_t mh, ml;
        if (a < 18446744073709551616.0) {
            mh = 0;
            ml = (uint64_t)a;  // truncates
        } else {  // an integer m * 2^e with e >= 12
            uint64_t b;
            memcpy(&b, &a, 8);
            const uint64_t m = (b & 0x000FFFFFFFFFFFFFull) | 0x0010000000000000ull;
            const int sh = (int)((b >> 52) & 0x7FFu) - 1075;  // 12 .. 74
            if (sh >= 64) {
                mh = m << (sh - 64);
                ml = 0;
            } else {
                mh = m >> (64 - sh);
                ml = m << sh;
            }
        }
        if (x < 0) {  // negate
            ml = ~ml + 1u;
            mh = ~mh + (ml == 0 ? 1u : 0u);
        }
        hi = mh;
        lo = ml;
    }
    // * 1_000_000 mod 2^128
    uint64_t ph, pl;
    shortest::mul64(lo, 1000000u, &ph, &pl);
    const uint64_t rh = ph + hi * 1000000u, rl = pl;
    // the valid range needs |ns| < 2^79: hi must be a sign extension with small magnitude
    const bool neg = (rh >> 63) != 0;
    uint64_t ah = rh, al = rl;
    if (neg) {
        al = ~al + 1u;
        ah = ~ah + (al == 0 ? 1u : 0u);
    }
    if (ah >= (1ull << 20)) return false;  // |ns| >= 2^84: far outside
    // |ns| = ah * 2^64 + al with ah < 2^20  ->  seconds and nanoseconds: long division by 10^9 in two 32-bit steps
    const uint64_t d = 1000000000ull;
    const uint64_t c1 = (ah << 32) | (al >> 32);  // < 2^52
    const uint64_t qh = c1 / d;                   // < 2^23
    const uint64_t c0 = ((c1 % d) << 32) | (al & 0xFFFFFFFFull);  // < 10^9 * 2^32
    const uint64_t q = (qh << 32) + c0 / d;       // c0 / d < 2^32
    const uint64_t rem = c0 % d;
    int64_t s = (int64_t)q;
    uint32_t ns = (uint32_t)rem;
    if (neg) {
        s = -s;
        if (ns) {
            s -= 1;
            ns = 1000000000u - ns;
        }
    }
    if (s < kMinUnix || s > kMaxUnix) return false;
    *secs = s;
    *nanos = ns;
    return true;
}
FGE_HD int64_t f64_as_i64(double x) {  // Rust `as i64`: truncating, saturating, NaN -> 0
    if (x != x) return 0;
    if (x >= 9223372036854775808.0) return INT64_MAX;
    if (x <= -9223372036854775808.0) return INT64_MIN;
    return (int64_t)x;
}
struct Civil {
    int y, m, d;
    uint32_t hh, mm, ss;
};
FGE_HD Civil civil_of(int64_t secs) {
    const int64_t days = secs >= 0 ? secs / 86400 : -((-secs + 86399) / 86400);
    const uint32_t sod = (uint32_t)(secs - days * 86400);
    Civil c;
    civil_from_days(days, &c.y, &c.m, &c.d);
    c.hh = sod / 3600u;
    c.mm = sod / 60u % 60u;
    c.ss = sod % 60u;
    return c;
}

template <class S, class R>
struct Rfc5424Emitter : Base<S, R> {
    using B = Base<S, R>;
    using B::li;
    using B::meta;
    using B::out;
    using B::t;
    FGE_HD Rfc5424Emitter(S& o, const EncCfg& c, R r, const DevTables& tb, uint64_t l, uint32_t m, const RowRegs* pre) : B{o, c, r, tb, l, m} { this->load_row(pre); }
    FGE_HD uint32_t run() {
        int64_t secs;
        uint32_t nanos;
        if (!rfc5424_ts_split(this->record_ts(), &secs, &nanos)) return ES_5424_DATE;
        const Civil c = civil_of(secs);
        if (c.y < 0 || c.y > 9999) return ES_5424_FORMAT;
        if (this->has_pri()) this->pri();
        else this->lit("<13>", 4);
        // "1 YYYY-MM-DDTHH:" (sixteen bytes) + "MM:SS" [+ "." + the nanoseconds without trailing zeros: time 0.3 Rfc3339] + "Z ": three
        // pieces (two without a fraction) instead of ~thirty one-byte puts
        {
            const uint32_t yh = B::d2((uint32_t)c.y / 100u), yl = B::d2((uint32_t)c.y % 100u), mo = B::d2((uint32_t)c.m), dd = B::d2((uint32_t)c.d);
            const uint32_t hh = B::d2(c.hh), mi = B::d2(c.mm), ss = B::d2(c.ss);
            out.put16((uint32_t)'1' | (uint32_t)' ' << 8 | yh << 16, yl | (uint32_t)'-' << 16 | (mo & 0xFFu) << 24,
                      (mo >> 8) | (uint32_t)'-' << 8 | dd << 16, (uint32_t)'T' | hh << 8 | (uint32_t)':' << 24);
            const uint32_t t0 = mi | (uint32_t)':' << 16 | (ss & 0xFFu) << 24, t1 = ss >> 8;  // "MM:SS"
            if (!nanos) {
                out.put_part(t0, t1 | (uint32_t)'Z' << 8 | (uint32_t)' ' << 16, 0u, 0u, 7u);
            } else {
                out.put_part(t0, t1, 0u, 0u, 5u);
                uint32_t digs = 9, v = nanos;
                while (v % 10u == 0) {
                    v /= 10u;
                    --digs;
                }
                // '.' + nine digits, cut behind `digs` of them, "Z " behind the cut
                const uint32_t lo8 = dtoa::Bcd17::bcd8(nanos % 100000000u);
                uint32_t q[4] = {(uint32_t)'.' | ((uint32_t)'0' + nanos / 100000000u) << 8, 0u, 0u, 0u};
                const uint32_t a = dtoa::ascii4(lo8 >> 16), b = dtoa::ascii4(lo8 & 0xFFFFu);  // digits 2..5, 6..9
                q[0] |= a << 16;
                q[1] = a >> 16 | b << 16;
                q[2] = b >> 16;
                B::keep_low(q, 1u + digs);
                B::set_byte(q, 1u + digs, (uint32_t)'Z');
                B::set_byte(q, 2u + digs, (uint32_t)' ');
                out.put_part(q[0], q[1], q[2], q[3], 3u + digs);
            }
        }
        if (this->some(S_HOST)) this->raw_field(S_HOST, (uint32_t)' ');
        else out.put(' ');
        if (this->some(S_APP)) this->raw_field(S_APP, (uint32_t)' ');
        if (this->some(S_PROC)) this->raw_field(S_PROC, (uint32_t)' ');
        else this->lit("- ", 2);
        if (this->some(S_MSGID)) this->raw_field(S_MSGID, (uint32_t)' ');
        else this->lit("- ", 2);
        if (this->has_sd()) {
            this->sd_display();
            out.put(' ');
        } else {
            this->lit("- ", 2);
        }
        if (this->some(S_MSG)) this->raw_field(S_MSG);
        return ES_OK;
    }
};

template <class S, class R>
struct Rfc3164Emitter : Base<S, R> {
    using B = Base<S, R>;
    using B::cfg;
    using B::out;
    FGE_HD Rfc3164Emitter(S& o, const EncCfg& c, R r, const DevTables& tb, uint64_t l, uint32_t m, const RowRegs* pre) : B{o, c, r, tb, l, m} { this->load_row(pre); }
    FGE_HD uint32_t run() {
        const int64_t secs = f64_as_i64(this->record_ts());
        if (secs < kMinUnix || secs > kMaxUnix) return ES_3164_TS;
        if (cfg.prepend_len != 0xFFFFFFFFu) this->blob(cfg.prepend_off, cfg.prepend_len);
        if (this->has_pri()) this->pri();
        const Civil c = civil_of(secs);
        // "[month repr:short]  [day padding:none] [hour]:[minute]:[second] "
        const char* mon = "JanFebMarAprMayJunJulAugSepOctNovDec" + 3 * (c.m - 1);
        {   // "Mon  " (two spaces) + the day without padding + " ", then "hh:mm:ss ": two pieces
            const uint32_t day = (uint32_t)c.d, nd = day >= 10u ? 2u : 1u;
            const uint32_t m4 = (uint32_t)(uint8_t)mon[0] | (uint32_t)(uint8_t)mon[1] << 8 | (uint32_t)(uint8_t)mon[2] << 16 | (uint32_t)' ' << 24;
            out.put_part(m4, nd == 2u ? ((uint32_t)' ' | B::d2(day) << 8 | (uint32_t)' ' << 24) : ((uint32_t)' ' | ((uint32_t)'0' + day) << 8 | (uint32_t)' ' << 16),
                         0u, 0u, 6u + nd);
            const uint32_t hh = B::d2(c.hh), mi = B::d2(c.mm), ss = B::d2(c.ss);
            out.put_part(hh | (uint32_t)':' << 16 | (mi & 0xFFu) << 24, (mi >> 8) | (uint32_t)':' << 8 | ss << 16, (uint32_t)' ', 0u, 9u);
        }
        if (this->some(S_HOST)) this->raw_field(S_HOST, (uint32_t)' ');
        else out.put(' ');
        if (this->some(S_APP)) this->raw_field(S_APP);
        if (this->some(S_PROC)) {
            out.put('[');
            this->raw_field(S_PROC, (uint32_t)']');
            this->lit(": ", 2);
        }
        if (this->some(S_MSGID)) this->raw_field(S_MSGID, (uint32_t)' ');
        if (this->has_sd()) {
            this->sd_display();
            out.put(' ');
        }
        if (this->some(S_MSG)) this->raw_field(S_MSG);
        return ES_OK;
    }
};

template <class S, class R>
struct PassthroughEmitter : Base<S, R> {
    using B = Base<S, R>;
    using B::cfg;
    FGE_HD PassthroughEmitter(S& o, const EncCfg& c, R r, const DevTables& tb, uint64_t l, uint32_t m, const RowRegs* pre) : B{o, c, r, tb, l, m} { this->load_row(pre); }
    FGE_HD uint32_t run() {
        if (!this->some(S_FULL)) return ES_PASSTHROUGH_EMPTY;
        if (cfg.prepend_len != 0xFFFFFFFFu) this->blob(cfg.prepend_off, cfg.prepend_len);
        this->raw_field(S_FULL);
        return ES_OK;
    }
};

// ---- mergers ------------------------------------------------------------------------------------
FGE_HD uint32_t dec_digits(uint64_t v) {
    uint32_t n = 1;
    while (v >= 10u) {
        v /= 10u;
        ++n;
    }
    return n;
}
// framed size of an encoded message of `n` bytes
FGE_HD uint64_t framed_size(uint32_t merger, uint64_t n) {
    switch (merger) {
        case FG_MERGE_LINE:
        case FG_MERGE_NUL: return n + 1u;
        case FG_MERGE_SYSLEN: return n + 2u + dec_digits(n + 1u);  // "{n+1} " + bytes + "\n"
        default: return n;
    }
}
// the encoded size of a syslen frame of `total` bytes (framed_size is strictly increasing: unique)
FGE_HD uint64_t syslen_payload(uint64_t total) {
    for (uint32_t d = 1; d <= 20u; ++d) {
        if (total < 2u + d) break;
        const uint64_t n = total - 2u - d;
        if (dec_digits(n + 1u) == d) return n;
    }
    return 0;
}


// ---- one row through encoder ENC (+ merger): the per-line body of the count and the write kernels, shared with the
//      host tests.  keys64 / slot_ent / order: kSortSlots scratch entries each (used by the GELF emitter only). ------
template <uint32_t ENC, class S, class R>
FGE_HD uint32_t encode_row(S& sink, const EncCfg& cfg, R rd, const DevTables& t, uint64_t li, uint32_t meta, uint64_t* keys64,
                           uint8_t* slot_ent, uint8_t* order, const RowRegs* pre = nullptr) {
    if (ENC == FG_ENC_GELF) {
        GelfEmitter<S, R> em(sink, cfg, rd, t, li, meta, pre);
        return em.run(keys64, slot_ent, order);
    } else if (ENC == FG_ENC_LTSV) {
        LtsvEmitter<S, R> em(sink, cfg, rd, t, li, meta, pre);
        return em.run();
    } else if (ENC == FG_ENC_RFC5424) {
        Rfc5424Emitter<S, R> em(sink, cfg, rd, t, li, meta, pre);
        return em.run();
    } else if (ENC == FG_ENC_RFC3164) {
        Rfc3164Emitter<S, R> em(sink, cfg, rd, t, li, meta, pre);
        return em.run();
    } else {
        PassthroughEmitter<S, R> em(sink, cfg, rd, t, li, meta, pre);
        return em.run();
    }
}
// count pass: the framed size of row li (0 when nothing is produced) and its encode status
template <uint32_t ENC, class R>
FGE_HD uint32_t row_size(const EncCfg& cfg, R rd, const DevTables& t, uint64_t li, uint32_t meta, uint64_t* keys64, uint8_t* slot_ent,
                         uint8_t* order, uint32_t* status, const RowRegs* pre = nullptr, uint32_t* plain = nullptr) {
    if (plain) *plain = 0u;
    if (FG_META_STATUS(meta) != 0u) {
        *status = ES_DECODE_FAILED;
        return 0;
    }
    CountSink cs;
    const uint32_t st = encode_row<ENC>(cs, cfg, rd, t, li, meta, keys64, slot_ent, order, pre);
    *status = st;
    if (plain) *plain = cs.slow ? 0u : 1u;  // (RowRegs::plain of the write pass)
    return st == ES_OK ? (uint32_t)framed_size(cfg.merger, cs.n) : 0u;
}
// write pass: `total` = the framed size the count pass returned for this row (the sink starts at the row's offset)
template <uint32_t ENC, class W, class R>
FGE_HD void row_write(W& sink, uint64_t total, const EncCfg& cfg, R rd, const DevTables& t, uint64_t li, uint32_t meta, uint64_t* keys64,
                      uint8_t* slot_ent, uint8_t* order, const RowRegs* pre = nullptr) {
    if (total == 0 || FG_META_STATUS(meta) != 0u) return;
    if (cfg.merger == FG_MERGE_SYSLEN) {  // "{len + 1} " in front (syslen_merger.rs:18-20)
        u64_digits(syslen_payload(total) + 1u, [&](uint32_t c) { sink.put(c); });
        sink.put(' ');
    }
    (void)encode_row<ENC>(sink, cfg, rd, t, li, meta, keys64, slot_ent, order, pre);
    if (cfg.merger == FG_MERGE_LINE || cfg.merger == FG_MERGE_SYSLEN) sink.put('\n');
    else if (cfg.merger == FG_MERGE_NUL) sink.put(0u);
    sink.finish();
}

}  // namespace emit
}  // namespace fg
