// fg_gelf2.hpp -- wave-cooperative tokeniser for GelfDecoder::decode
// (reference: src/flowgger/decoder/gelf_decoder.rs:34-125; JSON semantics = serde_json 0.8).
//
// The round-1 kernel gave every lane a LINE: a group cost max-over-lanes of the member count, the lanes diverged on the
// value kinds, and every member was a chain of dependent LDS round trips in ONE lane.  Here the work item is smaller than
// a line and the structural scan is byte-parallel, as the north_star describes (ballot / prefix-sum delimiter scans):
//
//   stage A   (pipeline, byte-parallel) five class bitmaps of the tile, one bit per byte:  '"' | '\' | {}[], | != ' ' | < 0x20
//   word pass (lane = 64 tile bytes)    escaped characters from the backslash runs, REAL quotes, the in-string mask as a
//                                       prefix-XOR of the real quotes carried across the wave (popcount parity + ballot), reset
//                                       at every line start; structural characters OUTSIDE strings -> the ITEM bitmap; a wave
//                                       prefix sum numbers the items
//   rows      (lane = one item)         the next lines of the tile take one ROW of W lanes each (W = 16: four lines per trip; 32 / 64 for
//                                       wider lines); an item is a '{' or ',' (it owns the member that follows) or the closing '}'.  The
//                                       lane delimits  ws "key" ws : ws value ws  with bit scans of 64-bit bitmap windows held in registers,
//                                       checks every byte outside the tokens, parses numbers (serde_json 0.8 algorithm) and literals.
//                                       What the lanes of a row find out about their LINE is combined by row reductions in registers
//                                       (DPP rotations): every lane of the row knows the verdict itself -- no LDS flags, no atomics, no
//                                       barrier in the loop.  BTreeMap order: rank = extras of the row with a smaller key (fifteen DPP
//                                       rotations of the keys' first four bytes; ties, duplicate keys and wider rows: the 64-bit keys by
//                                       shuffle); duplicates: the last one wins; gelf_decoder.rs:51-106 is evaluated per member, the FIRST
//                                       error in sorted order wins through an LDS min (rare).  Member records never leave the registers.
//   output                              rows from the lane that owns the line; extras go to first(line) + rank among the line's extras:
//                                       the lanes of a trip store into ONE contiguous stretch of the entry table, reserved out
//                                       of the wave's chunk (fg_wave.hpp wave_alloc; the reservation lives in registers across the tile)
//
// Exactness: this is a FAST FORM.  It accepts flat objects whose keys hold no escapes, with ' ' as the only whitespace
// between tokens, at most kMaxLineItems structural characters -- every GELF producer's output -- and proves every byte of
// the line to be part of that shape.  Anything else (nesting, escaped keys, TAB / CR / LF between tokens, raw control
// characters = the '\n' retry, syntax errors, two keys sharing 7 bytes, ...) is handed back untouched (`handled = false`) and
// takes the exact general form of fg_gelf.hip, which owns every error message of malformed input.
//
// Portable: compiled by hipcc for gfx950 and by g++ over the fiber emulation of a wave (fg_wave.hpp) for the CPU suite.
#pragma once
#include "fg_numfold.hpp"
#include "fg_numparse.hpp"
#include "fg_tables_view.hpp"
#include "fg_wave.hpp"

namespace fg {
namespace gelf2 {

// status codes == index into the reference's error strings (fg_error_string)
enum : uint32_t {
    G_OK = 0,
    G_JSON = 1,     // "Invalid GELF input, unable to parse as a JSON object"   :49
    G_EMPTY = 2,    // "Empty GELF input"                                       :50
    G_TS = 3,       // "Invalid GELF timestamp"                                 :53
    G_HOST = 4,     // "GELF host name must be a string"                        :58
    G_SHORT = 5,    // "GELF short message must be a string"                    :66
    G_FULL = 6,     // "GELF full message must be a string"                     :74
    G_VERSTR = 7,   // "GELF version must be a string"                          :78
    G_VER = 8,      // "Unsupported GELF version"                               :80
    G_LEVEL = 9,    // "Invalid severity level"                                 :83
    G_LEVEL7 = 10,  // "Invalid severity level (too high)"                      :85
    G_SDTYPE = 11,  // "Invalid value type in structured data"                  :97
    G_NOHOST = 12   // "Missing hostname"                                       :110
};
enum : uint32_t { V_STRING = 0, V_BOOL = 1, V_F64 = 2, V_I64 = 3, V_U64 = 4, V_NULL = 5 };  // == FG_T_*
enum : uint32_t { K_TS = 0, K_HOST = 1, K_SHORT = 2, K_FULL = 3, K_VERSION = 4, K_LEVEL = 5, K_OTHER = 6 };

// stage-A classes (one 16-bit mask per 16-byte chunk each)
enum : uint32_t { C_Q = 0, C_B = 1, C_ST = 2, C_NS = 3, kClasses = 4 };
// (control characters end the fast form wherever they stand: they are not a bitmap but one DIRTY bit per 64-byte word of the tile,
//  kept at the start of the extra block -- classify() returns the chunk's control mask as m[kClasses])
constexpr uint32_t kMaxLineItems = 64;   // structural characters of a line on the fast form ('{' + commas + '}')
constexpr uint32_t kLines = 64;

// term4: the frame terminator ('\n' or NUL in all four bytes) that separates the lines of the tile, wv::kNoTerm without framing: a
// terminator is not part of any line, so it does not make its word dirty (nor does the zero padding, wv::kPastSpan).
//   C_ST is '{' '}' ',' only: a '[' or ']' outside a string always lands where a key, a value or trailing space must be, and the
//   member fails there.  C_NS is "byte > 0x20": what it says about control characters is never used (dirty lines are not fast).
FG_WV void classify(uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3, uint32_t m[kClasses + 1], uint32_t term4 = wv::kNoTerm) {
    const uint32_t x[4] = {x0, x1, x2, x3};
    uint32_t q[4], b[4], st[4], ns[4], ct[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        q[k] = wv::eq_flags(x[k], 0x22222222u);
        b[k] = wv::eq_flags(x[k], 0x5C5C5C5Cu);
        st[k] = wv::eq_flags(x[k], 0x2C2C2C2Cu) | wv::eq_flags(x[k], 0x7B7B7B7Bu) | wv::eq_flags(x[k], 0x7D7D7D7Du);
        ns[k] = wv::gt_flags(x[k], 0x20u);
        ct[k] = wv::ctrl_flags(x[k]);
    }
    uint32_t dirty = ct[0] | ct[1] | ct[2] | ct[3];
    if (term4 != wv::kNoTerm && dirty) {  // (rare beyond the terminators themselves)
        if (term4 == wv::kPastSpan) {
            dirty = 0;  // the zeros behind the staged bytes are nobody's control characters
        } else {
            dirty = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) dirty |= ct[k] & ~wv::eq_flags(x[k], term4);
        }
    }
    m[C_Q] = wv::gather16(q[0], q[1], q[2], q[3]);
    m[C_B] = wv::gather16(b[0], b[1], b[2], b[3]);
    m[C_ST] = wv::gather16(st[0], st[1], st[2], st[3]);
    m[C_NS] = wv::gather16(ns[0], ns[1], ns[2], ns[3]);
    m[kClasses] = dirty;  // nonzero: the chunk holds a byte < 0x20
}

// ---------------------------------------------------------------------------------------------
// stage A's store of one 16-byte chunk: four 16-bit class masks, and the dirty bit of the chunk's word
FG_WV void store_classes(const uint32_t m[kClasses + 1], uint16_t* bm16, uint32_t chunk, uint32_t stride) {
#pragma unroll
    for (uint32_t c = 0; c < kClasses; ++c) bm16[c * stride + chunk] = (uint16_t)m[c];
    if (m[kClasses]) {  // rare
        uint32_t* dirty = reinterpret_cast<uint32_t*>(bm16 + kClasses * stride);  // = the first bytes of the extra block
        wv::lds_or(&dirty[chunk >> 7], 1u << ((chunk >> 2) & 31u));
    }
}
// The wave's LDS beyond the tile and the class bitmaps ("extra" block of the pipeline)
// ---------------------------------------------------------------------------------------------
struct Lds {
    wv::Bytes T;             // tile bytes
    uint32_t* bm[kClasses];  // class bitmaps, one bit per tile byte (C_Q becomes the REAL quotes, C_ST the item bitmap)
    uint8_t* wpar;           // [words + 1] string parity carried INTO each 64-byte word
    uint16_t* wcnt;          // [words + 1] items before each word
    uint16_t* items;         // [item_cap] tile position of every item of the tile
    uint32_t* dirty;         // [words / 32 + 1] bit w: word w of the tile holds a control character (set by stage A, cleared here)
    uint32_t *l_flags, *l_err, *l_cnt, *l_eoff, *l_sev;  // [lines] per line (l_eoff: the line's first entry slot)
    uint32_t* l_row;         // [lines][8]: ts lo, ts hi, host off, host len, msg off, msg len, full off, full len
    double* p10;             // [23] 10^0 .. 10^22 (exact): the number parser's divisors without a trip to global memory
    uint32_t* dw;            // [16 + 12] digit weights of a dword by its 4-bit digit mask (parse_num24), then 10^0 .. 10^8
    uint32_t* ent_state;     // the wave's entry-slot reservation (persists across tiles; set by the caller, wv::wave_alloc)
    uint32_t alloc_chunk;    // its reservation size
    uint32_t item_cap;
};
FG_WVH uint32_t up8(uint32_t v) { return (v + 7u) & ~7u; }
// items the array holds for a tile: one structural character per 12 bytes (GELF producers: one per ~20); a tile with more takes
// the general form
FG_WVH uint32_t item_cap_for(uint32_t tile_cap) { return up8(tile_cap / 12u); }  // (+ 8 spare slots in the array)
FG_WVH uint32_t dirty_bytes(uint32_t tile_cap) { return up8((tile_cap / 64u / 32u + 2u) * 4u); }
// bytes of the extra block
FG_WVH uint32_t extra_bytes(uint32_t tile_cap, uint32_t lines) {
    const uint32_t words = tile_cap / 64u + 2u;
    return dirty_bytes(tile_cap) + up8(words) + up8(words * 2u) + up8(item_cap_for(tile_cap) * 2u + 16u) + lines * (5u * 4u + 32u) +
           23u * 8u + 28u * 4u + 64u;
}
FG_WV Lds carve(const uint8_t* tile, uint16_t* bm16, uint32_t tile_cap, uint8_t* extra, uint32_t lines) {
    Lds L;
    L.T.w = reinterpret_cast<const uint32_t*>(tile);
    const uint32_t stride16 = tile_cap / 16u + 16u;
    for (uint32_t c = 0; c < kClasses; ++c) L.bm[c] = reinterpret_cast<uint32_t*>(bm16 + c * stride16);
    const uint32_t words = tile_cap / 64u + 2u;
    L.item_cap = item_cap_for(tile_cap);
    uint8_t* p = extra;
    L.dirty = reinterpret_cast<uint32_t*>(p); p += dirty_bytes(tile_cap);  // (first: stage A finds it right behind the bitmaps)
    L.p10 = reinterpret_cast<double*>(p); p += 23u * 8u;
    L.l_row = reinterpret_cast<uint32_t*>(p); p += lines * 32u;
    L.l_flags = reinterpret_cast<uint32_t*>(p); p += lines * 4u;
    L.l_err = reinterpret_cast<uint32_t*>(p); p += lines * 4u;
    L.l_cnt = reinterpret_cast<uint32_t*>(p); p += lines * 4u;
    L.l_eoff = reinterpret_cast<uint32_t*>(p); p += lines * 4u;
    L.l_sev = reinterpret_cast<uint32_t*>(p); p += lines * 4u;
    L.dw = reinterpret_cast<uint32_t*>(p); p += 28u * 4u;
    L.wcnt = reinterpret_cast<uint16_t*>(p); p += up8(words * 2u);
    L.items = reinterpret_cast<uint16_t*>(p); p += up8(L.item_cap * 2u + 16u);
    L.wpar = p;
    L.ent_state = nullptr;
    L.alloc_chunk = 64u;
    return L;
}
// once per wave, before the first tile: the powers of ten (exact doubles up to 10^22) and the digit-weight table
FG_WV void init_lds(const Lds& L) {
    numfold::init_tables(L.dw, L.p10);
    wv::sync();
}
// the dirty bits of a tile with room for tile_cap bytes (before the first tile; decode_tile clears what it has looked at)
FG_WV void clear_dirty(const Lds& L, uint32_t tile_cap) {
    for (uint32_t i = wv::lane(); i < dirty_bytes(tile_cap) / 4u; i += wv::kLanes) L.dirty[i] = 0u;
    wv::sync();
}

enum : uint32_t { LF_BAIL = 1, LF_CLOSED = 2, LF_HAVE_TS = 4, LF_HAVE_HOST = 8, LF_DUP = 16, LF_DONE = 32 };  // l_flags; FG_F_* in bits 8..15

// what the lane that owns a line gets back
struct LineOut {
    bool handled;  // false: the caller runs the exact general form for this line
    uint32_t status, severity, flags;
    uint32_t have_ts;
    double ts;
    uint32_t host_off, host_len, msg_off, msg_len, full_off, full_len;
    uint32_t first, n_ent;
};

// ---------------------------------------------------------------------------------------------
// small pieces
// ---------------------------------------------------------------------------------------------
FG_WV uint64_t prefix_xor64(uint64_t x) {
    x ^= x << 1;
    x ^= x << 2;
    x ^= x << 4;
    x ^= x << 8;
    x ^= x << 16;
    x ^= x << 32;
    return x;
}
// characters escaped by a backslash (simdjson's odd-backslash-run arithmetic); cin = bit 0 is escaped from the word before
FG_WV uint64_t find_escaped(uint64_t bs, uint64_t cin) {
    bs &= ~cin;
    const uint64_t follows = (bs << 1) | cin;
    const uint64_t even = 0x5555555555555555ull;
    const uint64_t odd_starts = bs & ~even & ~follows;
    const uint64_t seq_even = odd_starts + bs;
    const uint64_t invert = seq_even << 1;
    return (even ^ invert) & follows;
}
FG_WV uint64_t below(uint32_t bit) { return bit >= 64u ? ~0ull : (1ull << bit) - 1ull; }

FG_WV uint32_t known_key(uint32_t n, const uint32_t w[4]) {
    if (n == 9u && w[0] == 0x656D6974u && w[1] == 0x6D617473u && (w[2] & 0xFFu) == 'p') return K_TS;                            // timestamp
    if (n == 4u && w[0] == 0x74736F68u) return K_HOST;                                                                          // host
    if (n == 13u && w[0] == 0x726F6873u && w[1] == 0x656D5F74u && w[2] == 0x67617373u && (w[3] & 0xFFu) == 'e') return K_SHORT;  // short_message
    if (n == 12u && w[0] == 0x6C6C7566u && w[1] == 0x73656D5Fu && w[2] == 0x65676173u) return K_FULL;                           // full_message
    if (n == 7u && w[0] == 0x73726576u && (w[1] & 0xFFFFFFu) == 0x6E6F69u) return K_VERSION;                                    // version
    if (n == 5u && w[0] == 0x6576656Cu && (w[1] & 0xFFu) == 'l') return K_LEVEL;                                                // level
    return K_OTHER;
}
FG_WV bool hex4_ok(uint32_t v) {  // four ASCII hex digits in a dword
    const uint32_t d = v ^ 0x30303030u;
    const uint32_t dig_bad = (d | ((d & 0x7F7F7F7Fu) + 0x76767676u)) & 0x80808080u;
    const uint32_t l = (v | 0x20202020u) ^ 0x60606060u;
    const uint32_t let_hi = (l | ((l & 0x7F7F7F7Fu) + 0x79797979u)) & 0x80808080u;
    const uint32_t let_zero = ~(((l & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | l) & 0x80808080u;
    return (dig_bad & (let_hi | let_zero)) == 0u;
}
// JSON escapes of the string body [p, e) (serde_json 0.8 parse_escape): true = all valid.  Only called for bodies that hold
// a backslash; bmB = the backslash bitmap.
FG_WV bool escapes_ok(const wv::Bytes& T, const uint32_t* bmB, uint32_t p, uint32_t e) {
    for (;;) {
        const uint32_t h = wv::find_bit(bmB, p, e);
        if (h >= e) return true;
        if (h + 1u >= e) return false;
        const uint64_t v = T.load8(h);
        const uint32_t c = (uint32_t)(v >> 8) & 0xFFu;
        if (c == '"' || c == '\\' || c == '/' || c == 'b' || c == 'f' || c == 'n' || c == 'r' || c == 't') {
            p = h + 2u;
            continue;
        }
        if (c != 'u' || h + 6u > e) return false;
        const uint32_t hx = (uint32_t)(v >> 16);
        if (!hex4_ok(hx)) return false;
        const uint32_t d0 = hx & 0xFFu, d1 = (hx >> 8) & 0xFFu, d1l = d1 | 0x20u;
        const bool is_d = (d0 | 0x20u) == 'd';
        const bool high = is_d && (d1 == '8' || d1 == '9' || d1l == 'a' || d1l == 'b');
        const bool low = is_d && (d1l >= 'c' && d1l <= 'f');
        if (low) return false;  // lone low surrogate
        if (high) {             // must be followed by \uDC00..\uDFFF
            if (h + 12u > e) return false;
            const uint64_t v2 = T.load8(h + 6u);
            if (((uint32_t)v2 & 0xFFFFu) != (('u' << 8) | '\\')) return false;
            const uint32_t hx2 = (uint32_t)(v2 >> 16);
            if (!hex4_ok(hx2)) return false;
            const uint32_t e0 = hx2 & 0xFFu, e1 = ((hx2 >> 8) & 0xFFu) | 0x20u;
            if (!((e0 | 0x20u) == 'd' && e1 >= 'c' && e1 <= 'f')) return false;
            p = h + 12u;
        } else {
            p = h + 6u;
        }
    }
}
// reader over a token of the tile for fg_numparse (the rare shapes: exponents, 20+ digits): byte-wise out of LDS
struct TokReader {
    const wv::Bytes& T;
    FG_WV explicit TokReader(const wv::Bytes& t) : T(t) {}
    FG_WV uint32_t byte(uint32_t i) const { return T.byte(i); }
};
// serde_json 0.8 number (fg_numparse.hpp json_number) for the everyday shape  -?D+(.D+)?  with at most 19 digits, on a token
// of n <= 24 bytes held in registers: the significand by fg_numfold.hpp, then serde's arithmetic.  false = not that shape: the
// caller runs json_number, which owns every error.  dwt = Lds::dw.
FG_WV bool parse_num24(const uint32_t w[6], uint32_t n, const double* p10, const uint32_t* dwt, uint32_t* kind, uint64_t* bits) {
    const numfold::Folded f = numfold::fold24(w, n, dwt);
    if (!f.ok || (f.c0 == '0' && f.ni > 1u)) return false;  // (JSON: no leading zeros)
    const bool neg = f.neg, has_dot = f.has_dot;
    const uint32_t nf = f.nf;
    const uint64_t sig = f.sig;
    if (has_dot) {
        // visit_f64_from_parts: f = sig as f64; f /= POW10[nf]   (nf <= 18 here: the divisor comes from LDS)
        *kind = V_F64;
        const double f = (double)sig / p10[nf];
        *bits = num::f64_to_bits(neg ? -f : f);
        return true;
    }
    if (!neg) {
        *kind = V_U64;
        *bits = sig;
        return true;
    }
    const int64_t neg64 = (int64_t)(0ull - sig);
    if (neg64 > 0) {  // magnitude above i64: becomes a float
        *kind = V_F64;
        *bits = num::f64_to_bits(-(double)sig);
    } else if (neg64 < 0) {
        *kind = V_I64;
        *bits = (uint64_t)neg64;
    } else {
        *kind = V_U64;  // "-0" -> visit_i64(0) -> U64(0)
        *bits = 0;
    }
    return true;
}

// ---------------------------------------------------------------------------------------------
// The tile decoder.  Called by all 64 lanes in wave-uniform control flow.
//   span          staged tile bytes (wave-uniform)
//   valid         this lane owns a line that lies inside the tile: bytes [base, base + len)   (lane < lines of carve())
//   t             tables (entries of handled lines are written here)
// ---------------------------------------------------------------------------------------------
template <bool PROF = false>
FG_WV LineOut decode_tile(const Lds& L, uint32_t span, bool valid, uint32_t base, uint32_t len, const DevTables& t,
                          unsigned long long* phase = nullptr) {
    const uint32_t lane = wv::lane();
    // measurement build: cycles of  0 word pass  1 line pass  2 item parse (rest)  3 rank  4 dispatch  5 verdict + stash  6 copy-out + rows
    //                               7 item: fetch + line lookup  8 item: windows + delimiting  9 item: value
    uint64_t pc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tk = PROF ? wv::clock() : 0;
    auto tick = [&](int k) {
        if (PROF) {
            const uint64_t now = wv::clock();
            pc[k] += now - tk;
            tk = now;
        }
    };
    const wv::Bytes& T = L.T;
    uint64_t* Q = reinterpret_cast<uint64_t*>(L.bm[C_Q]);
    const uint64_t* B = reinterpret_cast<const uint64_t*>(L.bm[C_B]);
    uint64_t* ST = reinterpret_cast<uint64_t*>(L.bm[C_ST]);
    const uint32_t* bmQ = L.bm[C_Q];
    const uint32_t* bmB = L.bm[C_B];
    const uint32_t* bmN = L.bm[C_NS];
    LineOut out{};
    out.handled = false;

    // ================= word pass: real quotes, string parity, items =================
    const uint32_t nwords = (span + 63u) >> 6;
    const uint32_t wn = (nwords + wv::kLanes - 1u) / wv::kLanes;  // words per lane (wave-uniform)
    const uint32_t w0 = lane * wn < nwords ? lane * wn : nwords;
    const uint32_t w1 = w0 + wn < nwords ? w0 + wn : nwords;
    bool chain = false;  // a word of 64 backslashes: the escape carry would chain through it -- not worth a scan
    uint32_t par = 0;
    {
        uint32_t last_odd = 0;
        if (w1 > w0) {
            const uint64_t b = B[w1 - 1u];
            last_odd = b == ~0ull ? 0u : (wv::clz64(~b) & 1u);
        }
        uint32_t cin = wv::shfl_up1(last_odd, 0u);
        for (uint32_t w = w0; w < w1; ++w) {
            const uint64_t q = Q[w], b = B[w];
            chain = chain || b == ~0ull;
            const uint64_t qr = q & ~find_escaped(b, (uint64_t)cin);
            Q[w] = qr;
            L.wpar[w] = (uint8_t)par;
            par ^= wv::popc64(qr) & 1u;
            cin = b == ~0ull ? 0u : (wv::clz64(~b) & 1u);
        }
        const uint32_t pin = wv::mbcnt(wv::ballot(par != 0u)) & 1u;
        if (pin)
            for (uint32_t w = w0; w < w1; ++w) L.wpar[w] ^= 1u;
    }
    const bool tile_bail = wv::any(chain);
    wv::sync();
    // ---- the in-string state starts afresh at every line: toggle where the raw parity at a line's start differs from the line before
    bool use_t = false;
    uint32_t* Tg = reinterpret_cast<uint32_t*>(L.items);  // the toggle bitmap borrows the (not yet filled) item array
    {
        uint32_t rk = 0;
        if (valid) {
            const uint64_t qr = Q[base >> 6];
            rk = (uint32_t)L.wpar[base >> 6] ^ (wv::popc64(qr & below(base & 63u)) & 1u);
        }
        const uint32_t rprev = wv::shfl_up1(valid ? rk : 0u, 0u);
        const bool flip = valid && (rk ^ rprev) != 0u;
        use_t = wv::any(flip);
        if (use_t) {  // rare for GELF: a line with an odd number of quotes came before
            const uint32_t nw32 = nwords * 2u;
            for (uint32_t i = lane; i < nw32; i += wv::kLanes) Tg[i] = 0u;
            wv::sync();
            if (flip) wv::lds_xor(&Tg[base >> 5], 1u << (base & 31u));
            wv::sync();
            const uint64_t* Tg64 = reinterpret_cast<const uint64_t*>(Tg);
            par = 0;
            for (uint32_t w = w0; w < w1; ++w) {
                L.wpar[w] = (uint8_t)par;
                par ^= wv::popc64(Q[w] ^ Tg64[w]) & 1u;
            }
            const uint32_t pin = wv::mbcnt(wv::ballot(par != 0u)) & 1u;
            if (pin)
                for (uint32_t w = w0; w < w1; ++w) L.wpar[w] ^= 1u;
            wv::sync();
        }
    }
    // ---- items = structural characters outside strings
    uint32_t n_items_tile = 0;
    {
        const uint64_t* Tg64 = reinterpret_cast<const uint64_t*>(Tg);
        uint32_t cnt = 0;
        for (uint32_t w = w0; w < w1; ++w) {
            const uint64_t x = use_t ? (Q[w] ^ Tg64[w]) : Q[w];
            const uint64_t S = prefix_xor64(x) ^ (L.wpar[w] ? ~0ull : 0ull);
            const uint32_t left = span - 64u * w;
            const uint64_t E = ST[w] & ~S & below(left);
            ST[w] = E;
            L.wcnt[w] = (uint16_t)cnt;
            cnt += wv::popc64(E);
        }
        const uint32_t before = wv::excl_sum(cnt, &n_items_tile);
        if (use_t) wv::sync();  // (the item array is about to overwrite the toggle bitmap the loop above read)
        // ---- item positions, tile wide (lane = the word's owner) ----
        const bool fits = n_items_tile <= L.item_cap;  // wave-uniform
        uint32_t idx = before;
        for (uint32_t w = w0; w < w1; ++w) {
            L.wcnt[w] = (uint16_t)idx;
            uint64_t E = ST[w];
            while (E) {
                const uint32_t bit = wv::ctz64(E);
                E &= E - 1ull;
                if (fits) L.items[idx] = (uint16_t)(w * 64u + bit);
                ++idx;
            }
        }
        if (lane == 0) {
            L.wcnt[nwords] = (uint16_t)n_items_tile;
            ST[nwords] = 0ull;
        }
    }
    wv::sync();
    FG_MARK(0);
    tick(0);

    // ================= line pass =================
    const uint32_t s = base, e = base + len;
    uint32_t fi = 0, fe = 0;
    if (valid) {
        fi = (uint32_t)L.wcnt[s >> 6] + wv::popc64(ST[s >> 6] & below(s & 63u));
        fe = (uint32_t)L.wcnt[e >> 6] + wv::popc64(ST[e >> 6] & below(e & 63u));
    }
    // a control character in the line: not fast-form material.  The dirty bits say which 64-byte words hold one; a line that
    // touches a dirty word looks at its own bytes (rare: producers escape them)
    bool clean = true;
    if (valid && len != 0u && wv::any_bit(L.dirty, s >> 6, ((e - 1u) >> 6) + 1u)) {
        for (uint32_t i = s; i < e && clean; ++i) clean = L.T.byte(i) >= 0x20u;
    }
    const bool fast = valid && clean && !tile_bail && n_items_tile <= L.item_cap && len >= 2u && (fe - fi) >= 2u && (fe - fi) <= kMaxLineItems;
    if (valid) {
        L.l_flags[lane] = fast ? 0u : LF_BAIL;
        L.l_err[lane] = 0xFFFFFFFFu;
        L.l_cnt[lane] = 0u;
        L.l_eoff[lane] = 0u;
        L.l_sev[lane] = 0xFFu;
        uint32_t* row = L.l_row + lane * 8u;
        row[0] = row[1] = 0u;
        row[2] = 0u; row[3] = 0u;            // hostname: off, len
        row[4] = 0u; row[5] = FG_NONE;       // msg
        row[6] = 0u; row[7] = FG_NONE;       // full_msg
    }
    wv::sync();

    FG_MARK(1);

    tick(1);
    // ================= rows: the next lines of the tile, one line per row of W lanes, one item per lane =================
    // W = 16 (four lines at a time), 32 or 64: the narrowest row that holds the widest of the lines it would take.  What the lanes of
    // a row find out about their LINE (a member out of shape, the closing brace, the known keys met, the number of extras) is combined
    // by row reductions in registers, so every lane of the row knows the line's verdict itself: no LDS flags, no atomics and no
    // barrier inside the loop on the everyday path.  BTreeMap order among a line's extras: rank = extras of the row with a smaller
    // key, fifteen DPP rotations of the keys' first four bytes (W = 16); keys those cannot order, the same key twice, and wider
    // rows take the 64-bit form through LDS.
    const uint64_t valid_m = wv::ballot(valid);
    const uint32_t lend = valid_m ? 64u - wv::clz64(valid_m) : 0u;  // one past the last valid lane
    uint32_t lb = valid_m ? wv::ctz64(valid_m) : wv::kLanes;
    const uint32_t cntf = fast ? fe - fi : 0u;  // the items of this lane's line (0: not for the rows)
    const uint32_t pk_se = s | (e << 16), pk_fc = fi | (cntf << 16);
    // the wave's entry-slot reservation (wv::wave_alloc): read once per tile, kept in (scalar) registers, written back behind the loop
    uint32_t a_next = wv::bcast(L.ent_state[0], 0u), a_left = wv::bcast(L.ent_state[1], 0u);
    while (lb < lend) {
        const uint32_t c0 = wv::bcast(cntf, lb);
        const uint32_t c1 = lb + 1u < lend ? wv::bcast(cntf, lb + 1u) : 0u;
        const uint32_t c2 = lb + 2u < lend ? wv::bcast(cntf, lb + 2u) : 0u;
        const uint32_t c3 = lb + 3u < lend ? wv::bcast(cntf, lb + 3u) : 0u;
        const uint32_t m01 = c0 > c1 ? c0 : c1, m23 = c2 > c3 ? c2 : c3;
        const uint32_t lg = (m01 <= 16u && m23 <= 16u) ? 4u : (m01 <= 32u ? 5u : 6u);
        const uint32_t W = 1u << lg, rows = 64u >> lg;

        // ---- the item of this lane ----
        const uint32_t row = lane >> lg, j = lane & (W - 1u), k = lb + row;
        const uint32_t kk = k < lend ? k : lend - 1u;
        const uint32_t se = wv::shfl(pk_se, kk), fc = wv::shfl(pk_fc, kk);
        const uint32_t ls = se & 0xFFFFu, le_ = se >> 16;
        const uint32_t fi_k = fc & 0xFFFFu, n_k = k < lend ? fc >> 16 : 0u;  // the line's items: lanes [row base, row base + n_k)
        const bool act = j < n_k;
        FG_MARK(7);
        tick(7);
        bool member = false, closed = false, bad = false;
        uint64_t key = ~0ull, bits = 0;
        uint32_t key_b = 0, kl = 0, which = K_OTHER, kind = V_NULL, v_b = 0, v_len = 0, v_esc = 0;
        if (act) {
            const uint32_t pos = L.items[fi_k + j];
            const bool first_item = j == 0u, last_item = j + 1u == n_k;
            const uint32_t nxt = last_item ? pos + 1u : (uint32_t)L.items[fi_k + j + 1u];
            const uint32_t c = T.byte(pos), rb = T.byte(nxt);
            bool ok;
            if (c == '}') {
                // the closing brace: last item, only spaces behind it, and something opened before it
                ok = last_item && !first_item && wv::find_bit(bmN, pos + 1u, le_) >= le_;
                closed = ok;
            } else {
                // '{' or ',' owns the member up to the next item, which must be ',' or '}'  ('[' / ']' = nesting, control characters:
                // not fast-form material).  Straight-line from here: every read below is at a clamped, always readable position and the
                // verdict is ONE conjunction -- a short-circuit per test would cost a dozen scalar instructions of exec-mask work each.
                ok = (c == '{' || c == ',') & ((c == '{') == first_item) & !last_item & (rb == ',' || rb == '}');
                if (first_item && pos != ls) ok = ok & (wv::find_bit(bmN, ls, pos) >= pos);  // only spaces before the '{'
                // ---- the member's bytes (pos, nxt): 64-bit windows of the bitmaps from a = pos + 1 ----
                const uint32_t a = pos + 1u, m = nxt - a;
                const uint64_t in = below(m), top = 1ull << 63;
                const uint64_t NSr = wv::window64(bmN, a), NSw = NSr & in, Qw = wv::window64(bmQ, a) & in, Bw = wv::window64(bmB, a) & in;
                const bool empty = NSw == 0ull;
                // ws "key" ws : ws value ws   -- key, colon and the value's first byte inside the window
                const uint32_t p = wv::ctz64(NSw | top);
                const uint64_t above_p = ~1ull << p;
                const uint64_t q2 = Qw & above_p;
                const uint32_t ke = wv::ctz64(q2 | top);  // the key's closing quote
                const uint64_t above_ke = ~1ull << ke;
                const uint64_t n3 = NSw & above_ke;
                const uint32_t col = wv::ctz64(n3 | top);
                const uint64_t n4 = NSw & (~1ull << col);
                const uint32_t v = wv::ctz64(n4 | top);
                key_b = a + p + 1u;
                kl = ke - p - 1u;
                const bool okm = ((Qw >> p) & 1ull) & (q2 != 0ull) & ((Bw & above_p & below(ke)) == 0ull) & (n3 != 0ull) &
                                 (T.byte(a + col) == ':') & (n4 != 0ull) & (kl <= 255u);
                uint32_t vend = v;  // window-relative index just past the value
                bool okv = false;
                FG_MARK(8);
                tick(8);
                if (ok & okm & !empty) {
                    if ((Qw >> v) & 1ull) {
                        // ---- string: the closing quote is the next real quote (beyond the window: bit scan of the bitmap) ----
                        kind = V_STRING;
                        const uint64_t above_v = ~1ull << v;
                        const uint64_t q3 = Qw & above_v;
                        uint32_t ve;
                        bool esc;
                        uint64_t bsw = 0ull;  // the body's backslashes, when the whole body lies inside the first window
                        bool whole = false;
                        if (q3 != 0ull) {
                            ve = wv::ctz64(q3);
                            bsw = Bw & above_v & below(ve);
                            whole = true;
                            esc = bsw != 0ull;
                            okv = (NSw & (~1ull << ve)) == 0ull && (m <= 64u || wv::find_bit(bmN, a + 64u, nxt) >= nxt);
                        } else if (m > 64u && m <= 128u) {
                            // the closing quote in the second 64 bytes of the member (a short_message of 60+ bytes): one more window
                            const uint64_t in2 = below(m - 64u);
                            const uint64_t Q2 = wv::window64(bmQ, a + 64u) & in2, N2 = wv::window64(bmN, a + 64u) & in2;
                            const uint64_t B2 = wv::window64(bmB, a + 64u) & in2;
                            const uint32_t ve2 = wv::ctz64(Q2 | top);
                            ve = Q2 != 0ull ? 64u + ve2 : m;
                            esc = ((Bw & above_v) | (B2 & below(ve2))) != 0ull;
                            okv = Q2 != 0ull && (N2 & (~1ull << ve2)) == 0ull;
                        } else {
                            ve = m > 64u ? wv::find_bit(bmQ, a + 64u, nxt) - a : m;
                            esc = ve < m && wv::any_bit(bmB, a + v + 1u, a + ve);
                            okv = ve < m && wv::find_bit(bmN, a + ve + 1u, nxt) >= nxt;
                        }
                        v_b = a + v + 1u;
                        v_len = ve - v - 1u;
                        vend = ve + 1u;
                        if (okv && esc) {
                            v_esc = 1;
                            // the everyday case: ONE escape, of the two-byte kind
                            bool simple = false;
                            if (whole) {
                                const uint32_t hb = wv::ctz64(bsw | top);
                                const uint32_t c2 = T.byte(a + hb + 1u);
                                const bool two = c2 == '"' || c2 == '\\' || c2 == '/' || c2 == 'b' || c2 == 'f' || c2 == 'n' || c2 == 'r' || c2 == 't';
                                simple = two && hb + 1u < ve && (bsw & ~(3ull << hb)) == 0ull;
                            }
                            if (!simple) okv = escapes_ok(T, bmB, v_b, a + ve);
                        }
                    } else {
                        // ---- number / literal: the token ends at the first space (or at the delimiter) ----
                        const uint64_t sp = ~NSr & (~0ull << v);
                        const uint32_t te = wv::ctz64(sp | top);
                        vend = te < m ? te : m;
                        const uint32_t n = vend - v;
                        okv = m <= 64u && (NSw & ~below(vend)) == 0ull;  // (a token with 40 spaces behind it is not fast-form material)
                        uint32_t wv6[6];
                        T.load24(a + v, wv6);
                        const uint32_t cv = wv6[0] & 0xFFu;
                        if (cv == '-' || (cv - '0') <= 9u) {
                            uint32_t k2 = 0;
                            bool good = n <= 24u && parse_num24(wv6, n, L.p10, L.dw, &k2, &bits);
                            if (!good) {  // exponents, 20+ digits, malformed: the byte-wise parser decides
                                TokReader rd(T);
                                uint32_t endp = 0;
                                good = num::json_number(rd, a + v, a + vend, &endp, &k2, &bits) && endp == a + vend;
                            }
                            okv = okv && good;
                            kind = k2;
                        } else {
                            const bool tt = n == 4u && wv6[0] == 0x65757274u;                            // true
                            const bool ff2 = n == 5u && wv6[0] == 0x736C6166u && (wv6[1] & 0xFFu) == 'e';  // false
                            const bool nn = n == 4u && wv6[0] == 0x6C6C756Eu;                            // null
                            okv = okv && (tt || ff2 || nn);
                            kind = nn ? V_NULL : V_BOOL;
                            bits = tt ? 1u : 0u;
                        }
                    }
                    FG_MARK(9);
                    tick(9);
                    if (okv) {
                        uint32_t kw[4];
                        T.load16(key_b, kw);
                        which = known_key(kl, kw);
                        uint64_t pre = (uint64_t)kw[0] | ((uint64_t)kw[1] << 32);
                        if (kl < 8u) pre &= kl == 0u ? 0ull : (~0ull >> (64u - 8u * kl));
                        const uint64_t be = __builtin_bswap64(pre);
                        key = (be & ~0xFFull) | (which == K_OTHER ? 0x80ull : 0ull) | j;  // index in line < 64
                        member = true;
                    }
                }
                // "{}": no member at all; "{,", ",,", ",}" are syntax errors
                ok = ok & (empty ? (m <= 64u && c == '{' && rb == '}') : (okm & okv));
            }
            bad = !ok;
            member = member && ok;
        }
        FG_MARK(2);
        tick(2);

        // ---- gelf_decoder.rs:51-106 for this member: its status, the line flags it sets and what it leaves in the line's row ----
        uint32_t st = G_OK, fl = 0u, si = 0xFFu, pa = 0u, pb = 0u;
        if (member) {
            switch (which) {
                case K_TS: {
                    double tsv = 0.0;
                    if (kind == V_F64) tsv = num::bits_to_f64(bits);
                    else if (kind == V_U64) tsv = (double)bits;
                    else if (kind == V_I64) tsv = (double)(int64_t)bits;
                    else st = G_TS;
                    const uint64_t tb = num::f64_to_bits(tsv);
                    pa = (uint32_t)tb;
                    pb = (uint32_t)(tb >> 32);
                    si = 0u;
                    fl = LF_HAVE_TS;
                    break;
                }
                case K_HOST:
                    if (kind != V_STRING) st = G_HOST;
                    pa = v_b - ls;
                    pb = v_len;
                    si = 2u;
                    fl = LF_HAVE_HOST | (v_esc ? (uint32_t)FG_F_HOST_ESC << 8 : 0u);
                    break;
                case K_SHORT:
                    if (kind != V_STRING) st = G_SHORT;
                    pa = v_b - ls;
                    pb = v_len;
                    si = 4u;
                    fl = v_esc ? (uint32_t)FG_F_MSG_ESC << 8 : 0u;
                    break;
                case K_FULL:
                    if (kind != V_STRING) st = G_FULL;
                    pa = v_b - ls;
                    pb = v_len;
                    si = 6u;
                    fl = v_esc ? (uint32_t)FG_F_FULLMSG_ESC << 8 : 0u;
                    break;
                case K_VERSION:
                    if (kind != V_STRING) {
                        st = G_VERSTR;
                    } else if (v_esc) {
                        bad = true;  // "1.0" / "1.1" by DECODED value: an escaped spelling is for the general form
                    } else {
                        const uint32_t three = (uint32_t)T.load8(v_b) & 0xFFFFFFu;
                        if (!(v_len == 3u && (three == 0x302E31u || three == 0x312E31u))) st = G_VER;
                    }
                    break;
                case K_LEVEL:
                    if (kind != V_U64) st = G_LEVEL;  // Value::as_u64 (NumCast): floats and negatives -> None
                    else if (bits > 7u) st = G_LEVEL7;
                    else {
                        si = 8u;
                        pa = (uint32_t)bits;
                    }
                    break;
                default:
                    break;  // an extra (nested values never reach the fast form)
            }
        }
        FG_MARK(4);
        tick(4);

        // ---- BTreeMap order among the line's extras ----
        bool extra = member && which == K_OTHER;
        uint32_t rank_x = 0;
        const bool wide = lg != 4u;
        if (!wide) {
            const uint32_t xk = extra ? (uint32_t)(key >> 32) : 0xFFFFFFFFu;  // (valid UTF-8 holds no 0xFF: nothing ties with a non-extra)
            rank_x = wv::row16_count_less(xk);
        }
        // ---- the line, as every lane of its row sees it ----
        //   R_or : bit 0 a lane out of shape, 1 closed, 2 / 3 timestamp / host met, 8..15 FG_F_* escapes, 16 a member in error,
        //          17..22 the known keys met;   R_add: extras | known keys << 8 | sum of the extras' ranks << 16
        uint32_t R_or, R_add;
        auto combine = [&]() {
            const uint32_t known = (member && which != K_OTHER) ? 1u : 0u;
            const uint32_t w_or = (bad ? (uint32_t)LF_BAIL : 0u) | (closed ? (uint32_t)LF_CLOSED : 0u) | (member ? fl : 0u) |
                                  ((member && st != G_OK) ? 0x10000u : 0u) | (known ? (0x20000u << which) : 0u);
            const uint32_t w_add = (extra ? (1u | (rank_x << 16)) : 0u) | (known << 8);
            R_or = wv::rows_or(w_or, lg);
            R_add = wv::rows_add(w_add, lg);
        };
        combine();
        // Ranks by four key bytes are a permutation of 0 .. nx-1 exactly when those are pairwise different (a tie takes one from the
        // sum); a known key met twice shows as more known members than known-key bits.
        bool redo;
        {
            const uint32_t nx = R_add & 0xFFu, nk = (R_add >> 8) & 0xFFu;
            const bool tie = !wide && (R_add >> 16) != ((nx * (nx - 1u)) >> 1);
            const bool dupk = nk != wv::popc32((R_or >> 17) & 0x3Fu);
            redo = wide || wv::any(act && (tie || dupk));
        }
        if (redo) {  // rare: 64-bit keys (7 key bytes | extra bit | index), every lane meets the other lanes of its row by shuffle
            const uint32_t rb = lane - j, kinfo = key_b | (kl << 16);
            bool dropped = false, bail = false;
            rank_x = 0;
            {
                const uint64_t mine = member ? key : ~0ull;
                const uint32_t klo = (uint32_t)mine, khi = (uint32_t)(mine >> 32);
                for (uint32_t d = 1; d < W; ++d) {  // (wave-uniform)
                    const uint32_t src = rb + ((j + d) & (W - 1u));
                    const uint32_t olo = wv::shfl(klo, src), ohi = wv::shfl(khi, src), oi = wv::shfl(kinfo, src);
                    const uint64_t ko = ((uint64_t)ohi << 32) | olo;
                    if (member) {
                        rank_x += (ko < key && (ko & 0x80ull)) ? 1u : 0u;
                        if ((ko >> 8) == (key >> 8)) {
                            // same 7-byte prefix: the same key twice (the later one wins, BTreeMap::insert), or two keys the prefix
                            // cannot order (the general form sorts them)
                            const uint32_t ob = oi & 0xFFFFu, ol = oi >> 16;
                            bool same = ol == kl;
                            for (uint32_t q = 7; q < kl && same; ++q) same = T.byte(key_b + q) == T.byte(ob + q);
                            if (!same) bail = true;
                            else if (src > lane) dropped = true;
                        }
                    }
                }
            }
            if (bail) bad = true;
            member = member && !dropped;
            extra = extra && !dropped;
            if (wv::any(dropped)) {  // recount without the dropped duplicates
                const uint64_t mine = member ? key : ~0ull;
                const uint32_t klo = (uint32_t)mine, khi = (uint32_t)(mine >> 32);
                rank_x = 0;
                for (uint32_t d = 1; d < W; ++d) {
                    const uint32_t src = rb + ((j + d) & (W - 1u));
                    const uint32_t olo = wv::shfl(klo, src), ohi = wv::shfl(khi, src);
                    const uint64_t ko = ((uint64_t)ohi << 32) | olo;
                    rank_x += (member && ko < key && (ko & 0x80ull)) ? 1u : 0u;
                }
            }
            combine();
        }
        FG_MARK(3);
        tick(3);
        // ---- the line's verdict ----
        const bool line_ok = (R_or & (LF_BAIL | LF_CLOSED)) == LF_CLOSED;
        uint32_t status = G_OK;
        if (wv::any(act && line_ok && (R_or & 0x10000u))) {  // rare: the FIRST error in BTreeMap order is the line's
            const uint32_t rb = lane - j;
            const uint64_t mine = member ? key : ~0ull;
            const uint32_t klo = (uint32_t)mine, khi = (uint32_t)(mine >> 32);
            uint32_t rank_all = 0;
            for (uint32_t d = 1; d < W; ++d) {  // (wave-uniform)
                const uint32_t src = rb + ((j + d) & (W - 1u));
                const uint32_t olo = wv::shfl(klo, src), ohi = wv::shfl(khi, src);
                rank_all += (((uint64_t)ohi << 32) | olo) < mine ? 1u : 0u;
            }
            if (member && st != G_OK && line_ok) wv::lds_min(&L.l_err[k], (rank_all << 8) | st);
            wv::sync();
            if (act && (R_or & 0x10000u)) status = L.l_err[k] & 0xFFu;
            wv::sync();  // (read before the row's first lane replaces it with the final status)
        }
        if (status == G_OK && !(R_or & LF_HAVE_HOST)) status = G_NOHOST;  // :110
        const uint32_t n_ent = (line_ok && status == G_OK) ? (R_add & 0xFFu) : 0u;
        // ---- entry slots for the rows out of the wave's chunk: what is left of it takes the lines whose slices fit, whole; the
        //      rest opens the next chunk (the offsets are wave-uniform: scalar arithmetic on the rows' first lanes) ----
        const uint32_t e0 = wv::bcast(n_ent, 0u);
        const uint32_t e1 = rows >= 2u ? wv::bcast(n_ent, W) : 0u;
        const uint32_t e2 = rows == 4u ? wv::bcast(n_ent, 2u * W) : 0u;
        const uint32_t e3 = rows == 4u ? wv::bcast(n_ent, 3u * W) : 0u;
        const uint32_t o1 = e0, o2 = o1 + e1, o3 = o2 + e2, total = o3 + e3;
        wv::Slots es{a_next, total, 0u, false};
        if (total <= a_left) {  // everyday: out of what is left of the chunk -- scalar arithmetic on the state kept in registers
            a_next += total;
            a_left -= total;
        } else {
            uint32_t cut_at = total;
            if (e3 != 0u && o3 + e3 > a_left) cut_at = o3;
            if (e2 != 0u && o2 + e2 > a_left) cut_at = o2;
            if (e1 != 0u && o1 + e1 > a_left) cut_at = o1;
            if (e0 != 0u && e0 > a_left) cut_at = 0u;
            if (lane == 0u) {
                L.ent_state[0] = a_next;
                L.ent_state[1] = a_left;
            }
            wv::sync();
            es = wv::wave_alloc(glb(t.ent_used), t.ent_cap, L.ent_state, total, cut_at, L.alloc_chunk);
            a_next = wv::bcast(L.ent_state[0], 0u);
            a_left = wv::bcast(L.ent_state[1], 0u);
        }
        const uint32_t bl_off = row == 0u ? 0u : row == 1u ? o1 : row == 2u ? o2 : o3;
        const bool ov = es.overflow && n_ent != 0u && bl_off >= es.cut;
        const uint32_t slot0 = es.at(bl_off);
        if (act && line_ok) {
            if (j == 0u) {  // for the lane that owns the line (rows, below)
                L.l_err[k] = ov ? (uint32_t)FG_ST_OVERFLOW : status;
                L.l_cnt[k] = ov ? 0u : n_ent;
                L.l_eoff[k] = n_ent ? slot0 : 0u;
                L.l_flags[k] = LF_DONE | (R_or & (LF_HAVE_TS | 0xFF00u));
            }
            if (member && si != 0xFFu) {
                if (si == 8u) {
                    L.l_sev[k] = pa;
                } else {
                    uint32_t* rowp = L.l_row + k * 8u;
                    rowp[si] = pa;
                    rowp[si + 1u] = pb;
                }
            }
            // ---- extras -> the entry table, at first(line) + rank among the line's extras (BTreeMap order): the lanes of a block
            //      write ONE contiguous stretch of slots ----
            if (extra && n_ent != 0u && !ov) {
                const uint32_t slot = slot0 + rank_x;
                gstore(t.ent_name, slot, fg_span{key_b - ls, kl});
                gstore(t.ent_val, slot, kind == V_STRING ? ((uint64_t)(v_b - ls) | ((uint64_t)v_len << 32)) : kind == V_NULL ? 0ull : bits);
                gstore(t.ent_type, slot, (uint8_t)kind);
                gstore(t.ent_flags, slot, (uint8_t)((kind == V_STRING && v_esc) ? FG_EF_VAL_ESC : 0));
            }
        }
        lb += rows;
        FG_MARK(5);
        tick(5);
    }
    if (lane == 0u) {
        L.ent_state[0] = a_next;
        L.ent_state[1] = a_left;
    }
    // the dirty bits of this tile have been looked at: clean for the next one
    if (wv::any(valid)) {
        for (uint32_t i = lane; i < (nwords + 31u) / 32u + 1u; i += wv::kLanes) L.dirty[i] = 0u;
    }
    wv::sync();
    // ================= rows =================
    if (valid && fast) {
        const uint32_t lf = L.l_flags[lane];
        if ((lf & (LF_BAIL | LF_DONE)) == LF_DONE) {
            out.handled = true;
            out.status = L.l_err[lane];
            out.n_ent = L.l_cnt[lane];
            out.first = out.n_ent ? L.l_eoff[lane] : 0u;
            out.have_ts = (lf & LF_HAVE_TS) ? 1u : 0u;
            out.flags = ((lf >> 8) & 0xFFu) | ((lf & LF_HAVE_TS) ? 0u : (uint32_t)FG_F_TS_NOW);  // :109
            out.severity = L.l_sev[lane];
            const uint32_t* row = L.l_row + lane * 8u;
            out.ts = num::bits_to_f64((uint64_t)row[0] | ((uint64_t)row[1] << 32));
            out.host_off = row[2]; out.host_len = row[3];
            out.msg_off = row[4]; out.msg_len = row[5];
            out.full_off = row[6]; out.full_len = row[7];
        }
    }
    FG_MARK(6);
    tick(6);
    if (PROF && phase && lane == 0)
        for (int k = 0; k < 10; ++k) phase[k] += (unsigned long long)pc[k];  // (the wave's own accumulators)
    return out;
}

}  // namespace gelf2
}  // namespace fg
