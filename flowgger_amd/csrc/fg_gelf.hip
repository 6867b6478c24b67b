// fg_gelf.hip -- gfx950 kernel for GelfDecoder::decode
// (reference: src/flowgger/decoder/gelf_decoder.rs:34-125; JSON semantics = serde_json 0.8).
//
// Runs on the streaming pipeline of fg_pipeline.hpp (persistent waves, register prefetch window,
// LDS tile per line group, at most 32 lines per group -- the per-lane member arrays live in LDS).
//
// FAST FORM (GelfFormat::fast_walk, the bottom half of this file): a flat object of at most 16
// members whose keys hold no escapes -- every GELF producer's output.  Stage A builds the bitmap
// of { '"', '\', bytes < 0x20 }; a lane walks its line member by member: strings are skipped with
// one bit scan of that bitmap per segment (escapes are validated in place), the punctuation
// between tokens is resolved in 16-byte register windows, numbers use serde_json 0.8's
// algorithm.  Every member becomes a 64-bit record + a sort key (first 7 key bytes, big endian,
// then the member index) in the lane's LDS array; the keys are ranked in registers (BTreeMap
// order, last duplicate wins), gelf_decoder.rs:51-106 is dispatched in that order, and the
// extras are copied to the entry table in the same order.  Anything else (nesting, escaped
// keys, raw control characters = the '\n' retry, syntax errors, 17+ members, two different keys
// sharing their first 7 bytes) leaves the fast form BEFORE any output and takes the exact
// general form below -- which is the original three-pass byte-walking implementation:
//   pass 1  strict JSON validation in document order (serde_json's grammar: whitespace set,
//           one leading zero, escapes, \uXXXX surrogate pairing, raw control characters in
//           strings = InvalidUnicodeCodePoint, arbitrary nesting with an explicit LDS bit-stack),
//           recording where each top-level member's key starts.  If the FIRST error is a raw
//           control character, the line is re-validated in "retry" mode = the reference's
//           `line.replace('\n', "\\n")` (gelf_decoder.rs:44-46) applied on the fly.
//   pass 2  serde_json 0.8's Value::Object is a BTreeMap: members are visited in byte order of
//           their DECODED keys, the last duplicate wins.  The lane sorts its member index
//           (insertion sort in LDS, <= 32 members) or, for larger objects, runs selection rounds
//           that re-walk the line; the dispatch of gelf_decoder.rs:51-106 then happens in that
//           order, so the FIRST error in sorted-key order is the one reported.
//   pass 3  extras are written to the entry table in sorted order (count -> wave-aggregated
//           atomic -> fill).
// Numbers use serde_json 0.8's own (not correctly rounded) algorithm, see fg_numparse.hpp.
#include "fg_pipeline.hpp"
#include "fg_numparse.hpp"

namespace fg {

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

enum : uint32_t { J_OK = 0, J_SYNTAX = 1, J_CTRL = 2 };
enum : uint32_t { V_STRING = 0, V_BOOL = 1, V_F64 = 2, V_I64 = 3, V_U64 = 4, V_NULL = 5, V_NESTED = 6 };
constexpr uint32_t kMaxDepth = 512;     // must equal the oracle's bound (serde_json 0.8 has none)
constexpr uint32_t kMaxStored = 32;     // members whose key position is kept in LDS

struct Member {
    uint32_t key_b, key_e, key_esc;  // key bytes between the quotes
    uint32_t kind;
    uint32_t v_b, v_e, v_esc;        // strings: bytes between the quotes
    uint64_t bits;                   // numbers / bool
    uint32_t end;                    // index just past the value
};

template <class R>
struct Gelf {
    R& rd;
    uint32_t len;
    bool retry;
    uint32_t* stack;  // kMaxDepth bits (LDS, per lane)

    __device__ __forceinline__ bool is_ws(uint32_t c) const {
        return c == ' ' || c == '\t' || c == '\r' || (c == '\n' && !retry);
    }
    __device__ __forceinline__ uint32_t skip_ws(uint32_t p) const {
        while (p < len && is_ws(rd.byte(p))) ++p;
        return p;
    }
    static __device__ __forceinline__ int hexv(uint32_t c) {
        if (c - '0' <= 9u) return (int)(c - '0');
        uint32_t l = c | 0x20u;
        if (l - 'a' <= 5u) return (int)(l - 'a' + 10);
        return -1;
    }
    __device__ __forceinline__ uint32_t hex4(uint32_t& p, uint32_t* out) const {
        uint32_t n = 0;
        for (int k = 0; k < 4; ++k) {
            if (p >= len) return J_SYNTAX;
            int h = hexv(rd.byte(p++));
            if (h < 0) return J_SYNTAX;
            n = n * 16u + (uint32_t)h;
        }
        *out = n;
        return J_OK;
    }
    // p = index just past the opening quote; on success *end = index of the closing quote.
    __device__ uint32_t scan_string(uint32_t p, uint32_t* end, uint32_t* esc) const {
        uint32_t has_esc = 0;
        for (;;) {
            if (p >= len) return J_SYNTAX;  // EOFWhileParsingString
            uint32_t c = rd.byte(p);
            if (c == '"') {
                *end = p;
                *esc = has_esc;
                return J_OK;
            }
            if (c == '\\') {
                has_esc = 1;
                ++p;
                if (p >= len) return J_SYNTAX;
                uint32_t e = rd.byte(p);
                if (retry && e == '\n') {
                    // "\<LF>" became "\\" "\n"... i.e. an escaped backslash followed by 'n'
                    ++p;
                    continue;
                }
                ++p;
                if (e == 'u') {
                    uint32_t n1;
                    if (hex4(p, &n1)) return J_SYNTAX;
                    if (n1 >= 0xDC00u && n1 <= 0xDFFFu) return J_SYNTAX;
                    if (n1 >= 0xD800u && n1 <= 0xDBFFu) {
                        if (p + 1 >= len) return J_SYNTAX;
                        if (rd.byte(p) != '\\' || rd.byte(p + 1) != 'u') return J_SYNTAX;
                        p += 2;
                        uint32_t n2;
                        if (hex4(p, &n2)) return J_SYNTAX;
                        if (n2 < 0xDC00u || n2 > 0xDFFFu) return J_SYNTAX;
                    }
                } else if (!(e == '"' || e == '\\' || e == '/' || e == 'b' || e == 'f' || e == 'n' || e == 'r' || e == 't')) {
                    return J_SYNTAX;  // InvalidEscape
                }
                continue;
            }
            if (c < 0x20u) {
                if (retry && c == '\n') {  // "\n" escape after the replace: fine (marks the span as escaped-free raw LF)
                    ++p;
                    continue;
                }
                return J_CTRL;  // InvalidUnicodeCodePoint
            }
            ++p;
        }
    }
    __device__ __forceinline__ bool lit(uint32_t p, const char* s, uint32_t n) const {
        if (p + n > len) return false;
        for (uint32_t i = 0; i < n; ++i)
            if (rd.byte(p + i) != (uint32_t)(uint8_t)s[i]) return false;
        return true;
    }
    // scalar value starting at p (already past whitespace, p < len, not a container)
    __device__ uint32_t scalar(uint32_t p, Member* m) const {
        uint32_t c = rd.byte(p);
        if (c == '"') {
            m->kind = V_STRING;
            m->v_b = p + 1;
            uint32_t e;
            if (uint32_t r = scan_string(p + 1, &e, &m->v_esc)) return r;
            m->v_e = e;
            m->end = e + 1;
            return J_OK;
        }
        if (c == 'n') {
            if (!lit(p + 1, "ull", 3)) return J_SYNTAX;
            m->kind = V_NULL;
            m->end = p + 4;
            return J_OK;
        }
        if (c == 't') {
            if (!lit(p + 1, "rue", 3)) return J_SYNTAX;
            m->kind = V_BOOL;
            m->bits = 1;
            m->end = p + 4;
            return J_OK;
        }
        if (c == 'f') {
            if (!lit(p + 1, "alse", 4)) return J_SYNTAX;
            m->kind = V_BOOL;
            m->bits = 0;
            m->end = p + 5;
            return J_OK;
        }
        if (c == '-' || (c - '0') <= 9u) {
            uint32_t end, kind;
            uint64_t bits;
            if (!num::json_number(rd, p, len, &end, &kind, &bits)) return J_SYNTAX;
            m->kind = kind;  // FG_T_F64/I64/U64 == V_F64/I64/U64
            m->bits = bits;
            m->end = end;
            return J_OK;
        }
        return J_SYNTAX;  // ExpectedSomeValue
    }
    // nested array/object starting at p ('[' or '{'): validate, return the index past its end
    __device__ uint32_t nested(uint32_t p, uint32_t depth0, uint32_t* end) const {
        uint32_t depth = depth0;
        bool first = true;
        // push
        {
            bool is_obj = rd.byte(p) == '{';
            if (++depth > kMaxDepth) return J_SYNTAX;
            uint32_t bit = depth - 1;
            if (is_obj) stack[bit >> 5] |= 1u << (bit & 31u);
            else stack[bit >> 5] &= ~(1u << (bit & 31u));
            ++p;
        }
        for (;;) {
            p = skip_ws(p);
            if (p >= len) return J_SYNTAX;
            const uint32_t top = depth - 1;
            const bool in_obj = (stack[top >> 5] >> (top & 31u)) & 1u;
            uint32_t c = rd.byte(p);
            if (c == (in_obj ? '}' : ']')) {
                ++p;
                if (--depth == depth0) {
                    *end = p;
                    return J_OK;
                }
                first = false;
                continue;
            }
            if (!first) {
                if (c != ',') return J_SYNTAX;
                ++p;
                if (in_obj) p = skip_ws(p);
            }
            first = false;
            if (in_obj) {
                if (p >= len || rd.byte(p) != '"') return J_SYNTAX;  // KeyMustBeAString
                uint32_t e, esc;
                if (uint32_t r = scan_string(p + 1, &e, &esc)) return r;
                p = skip_ws(e + 1);
                if (p >= len || rd.byte(p) != ':') return J_SYNTAX;
                ++p;
            }
            p = skip_ws(p);
            if (p >= len) return J_SYNTAX;
            c = rd.byte(p);
            if (c == '[' || c == '{') {
                if (++depth > kMaxDepth) return J_SYNTAX;
                uint32_t bit = depth - 1;
                if (c == '{') stack[bit >> 5] |= 1u << (bit & 31u);
                else stack[bit >> 5] &= ~(1u << (bit & 31u));
                ++p;
                first = true;
                continue;
            }
            Member tmp;
            if (uint32_t r = scalar(p, &tmp)) return r;
            p = tmp.end;
        }
    }
    // any value at p (past whitespace): scalar or container (validated, kind = V_NESTED)
    __device__ uint32_t value(uint32_t p, uint32_t depth, Member* m) const {
        if (p >= len) return J_SYNTAX;
        uint32_t c = rd.byte(p);
        if (c == '[' || c == '{') {
            m->kind = V_NESTED;
            return nested(p, depth, &m->end);
        }
        return scalar(p, m);
    }
    // top-level object member whose key's opening quote is at p
    __device__ uint32_t member(uint32_t p, Member* m) const {
        m->key_b = p + 1;
        uint32_t e;
        if (uint32_t r = scan_string(p + 1, &e, &m->key_esc)) return r;
        m->key_e = e;
        p = skip_ws(e + 1);
        if (p >= len || rd.byte(p) != ':') return J_SYNTAX;
        p = skip_ws(p + 1);
        return value(p, 1, m);
    }

    // ---- decoded-key byte stream (for ordering and matching keys that contain escapes) ----
    struct KeyIter {
        uint32_t p, e;
        uint32_t pend;   // up to 3 pending bytes, low byte first
        uint32_t npend;
    };
    __device__ int key_next(KeyIter& it) const {  // next decoded byte or -1
        if (it.npend) {
            int b = (int)(it.pend & 0xFFu);
            it.pend >>= 8;
            --it.npend;
            return b;
        }
        if (it.p >= it.e) return -1;
        uint32_t c = rd.byte(it.p++);
        if (c != '\\') return (int)c;
        uint32_t x = rd.byte(it.p);
        if (retry && x == '\n') {  // escaped backslash, then a literal 'n'
            ++it.p;
            it.pend = 'n';
            it.npend = 1;
            return '\\';
        }
        ++it.p;
        switch (x) {
            case 'b': return 8;
            case 'f': return 12;
            case 'n': return 10;
            case 'r': return 13;
            case 't': return 9;
            case 'u': {
                uint32_t q = it.p, n1 = 0, n2 = 0;
                hex4(q, &n1);
                if (n1 >= 0xD800u && n1 <= 0xDBFFu) {
                    q += 2;
                    hex4(q, &n2);
                    n1 = (((n1 - 0xD800u) << 10) | (n2 - 0xDC00u)) + 0x10000u;
                }
                it.p = q;
                if (n1 < 0x80u) return (int)n1;
                if (n1 < 0x800u) {
                    it.pend = 0x80u | (n1 & 0x3Fu);
                    it.npend = 1;
                    return (int)(0xC0u | (n1 >> 6));
                }
                if (n1 < 0x10000u) {
                    it.pend = (0x80u | ((n1 >> 6) & 0x3Fu)) | ((0x80u | (n1 & 0x3Fu)) << 8);
                    it.npend = 2;
                    return (int)(0xE0u | (n1 >> 12));
                }
                it.pend = (0x80u | ((n1 >> 12) & 0x3Fu)) | ((0x80u | ((n1 >> 6) & 0x3Fu)) << 8) | ((0x80u | (n1 & 0x3Fu)) << 16);
                it.npend = 3;
                return (int)(0xF0u | (n1 >> 18));
            }
            default: return (int)x;  // " \ /
        }
    }
    // three-way compare of two keys by decoded bytes (String Ord = byte order, shorter first)
    __device__ int key_cmp(uint32_t ab, uint32_t ae, uint32_t aesc, uint32_t bb, uint32_t be, uint32_t besc) const {
        if (!aesc && !besc) {
            uint32_t la = ae - ab, lb = be - bb, n = la < lb ? la : lb;
            for (uint32_t i = 0; i < n; ++i) {
                uint32_t x = rd.byte(ab + i), y = rd.byte(bb + i);
                if (x != y) return x < y ? -1 : 1;
            }
            return la == lb ? 0 : (la < lb ? -1 : 1);
        }
        KeyIter a{ab, ae, 0, 0}, b{bb, be, 0, 0};
        for (;;) {
            int x = key_next(a), y = key_next(b);
            if (x != y) return x < y ? -1 : 1;  // -1 (end) sorts first
            if (x < 0) return 0;
        }
    }
    __device__ bool key_is(uint32_t kb, uint32_t ke, uint32_t kesc, const char* s, uint32_t n) const {
        if (!kesc) return num::bytes_equal(rd, kb, ke, s, n);
        KeyIter a{kb, ke, 0, 0};
        for (uint32_t i = 0; i < n; ++i)
            if (key_next(a) != (int)(uint8_t)s[i]) return false;
        return key_next(a) < 0;
    }
};

struct GRow {
    uint32_t status = G_OK;
    uint32_t severity = 0xFF, flags = 0;
    double ts = 0.0;
    uint32_t have_ts = 0, have_host = 0;
    uint32_t host_off = 0, host_len = 0, msg_off = 0, msg_len = FG_NONE, full_off = 0, full_len = FG_NONE;
    uint32_t n_members = 0, n_ent = 0, stored = 0;
};

// gelf_decoder.rs:51-106 for ONE member (already the winner among duplicates).  EMIT writes
// extras to slot+*cnt.  Returns a G_* status (G_OK to continue).
template <bool EMIT, class R>
__device__ uint32_t gelf_dispatch(const Gelf<R>& g, const Member& m, GRow& r, const DevTables& t, uint32_t slot,
                                  uint32_t* cnt) {
    const uint32_t kb = m.key_b, ke = m.key_e, kx = m.key_esc;
    if (g.key_is(kb, ke, kx, "timestamp", 9)) {
        if (m.kind == V_F64) r.ts = num::bits_to_f64(m.bits);
        else if (m.kind == V_U64) r.ts = (double)m.bits;
        else if (m.kind == V_I64) r.ts = (double)(int64_t)m.bits;
        else return G_TS;
        r.have_ts = 1;
    } else if (g.key_is(kb, ke, kx, "host", 4)) {
        if (m.kind != V_STRING) return G_HOST;
        r.host_off = m.v_b;
        r.host_len = m.v_e - m.v_b;
        r.have_host = 1;
        if (m.v_esc) r.flags |= FG_F_HOST_ESC;
    } else if (g.key_is(kb, ke, kx, "short_message", 13)) {
        if (m.kind != V_STRING) return G_SHORT;
        r.msg_off = m.v_b;
        r.msg_len = m.v_e - m.v_b;
        if (m.v_esc) r.flags |= FG_F_MSG_ESC;
    } else if (g.key_is(kb, ke, kx, "full_message", 12)) {
        if (m.kind != V_STRING) return G_FULL;
        r.full_off = m.v_b;
        r.full_len = m.v_e - m.v_b;
        if (m.v_esc) r.flags |= FG_F_FULLMSG_ESC;
    } else if (g.key_is(kb, ke, kx, "version", 7)) {
        if (m.kind != V_STRING) return G_VERSTR;
        if (!g.key_is(m.v_b, m.v_e, m.v_esc, "1.0", 3) && !g.key_is(m.v_b, m.v_e, m.v_esc, "1.1", 3)) return G_VER;
    } else if (g.key_is(kb, ke, kx, "level", 5)) {
        if (m.kind != V_U64) return G_LEVEL;  // Value::as_u64 (NumCast): floats and negatives -> None
        if (m.bits > 7) return G_LEVEL7;
        r.severity = (uint32_t)m.bits;
    } else {
        if (m.kind == V_NESTED) return G_SDTYPE;
        if (EMIT) {
            const uint32_t k = slot + *cnt;
            t.ent_name[k] = fg_span{kb, ke - kb};
            t.ent_type[k] = (uint8_t)m.kind;
            uint32_t fl = kx ? FG_EF_NAME_ESC : 0;
            if (m.kind == V_STRING) {
                t.ent_val[k] = (uint64_t)m.v_b | ((uint64_t)(m.v_e - m.v_b) << 32);
                if (m.v_esc) fl |= FG_EF_VAL_ESC;
            } else {
                t.ent_val[k] = (m.kind == V_NULL) ? 0ull : m.bits;
            }
            t.ent_flags[k] = (uint8_t)fl;
        }
        ++*cnt;
    }
    return G_OK;
}

// pass 1: validate the document, count top-level members, remember key positions.
template <class R>
__device__ uint32_t gelf_validate(const Gelf<R>& g, uint32_t* keypos, GRow& r, bool* is_object) {
    uint32_t p = g.skip_ws(0);
    if (p >= g.len) return J_SYNTAX;  // EOFWhileParsingValue
    *is_object = g.rd.byte(p) == '{';
    Member m;
    if (!*is_object) {
        if (uint32_t e = g.value(p, 0, &m)) return e;
        p = g.skip_ws(m.end);
        return p == g.len ? J_OK : J_SYNTAX;
    }
    ++p;
    uint32_t n = 0, stored = 1;
    bool first = true;
    for (;;) {
        p = g.skip_ws(p);
        if (p >= g.len) return J_SYNTAX;
        uint32_t c = g.rd.byte(p);
        if (c == '}') {
            ++p;
            break;
        }
        if (!first) {
            if (c != ',') return J_SYNTAX;
            p = g.skip_ws(p + 1);
        }
        first = false;
        if (p >= g.len || g.rd.byte(p) != '"') return J_SYNTAX;
        if (n < kMaxStored && p < 65536u) keypos[n] = p;
        else stored = 0;
        if (uint32_t e = g.member(p, &m)) return e;
        p = m.end;
        ++n;
    }
    p = g.skip_ws(p);
    if (p != g.len) return J_SYNTAX;  // TrailingCharacters
    r.n_members = n;
    r.stored = stored;
    return J_OK;
}

// Visit the members in sorted decoded-key order, last duplicate wins; dispatch each.
template <bool EMIT, class R>
__device__ void gelf_sorted_dispatch(const Gelf<R>& g, uint32_t* keypos, GRow& r, const DevTables& t, uint32_t slot) {
    uint32_t cnt = 0;
    const uint32_t n = r.n_members;
    if (r.stored) {
        if (!EMIT) {
            // stable insertion sort of the key positions by decoded key (LDS, n <= 32)
            for (uint32_t i = 1; i < n; ++i) {
                uint32_t pi = keypos[i];
                uint32_t ie, iesc;
                g.scan_string(pi + 1, &ie, &iesc);
                uint32_t j = i;
                while (j > 0) {
                    uint32_t pj = keypos[j - 1];
                    uint32_t je, jesc;
                    g.scan_string(pj + 1, &je, &jesc);
                    if (g.key_cmp(pj + 1, je, jesc, pi + 1, ie, iesc) <= 0) break;
                    keypos[j] = pj;
                    --j;
                }
                keypos[j] = pi;
            }
        }
        for (uint32_t i = 0; i < n; ++i) {
            Member m;
            g.member(keypos[i], &m);
            if (i + 1 < n) {  // an equal key follows (stable sort => it is a later duplicate): skip
                uint32_t ne, nesc;
                g.scan_string(keypos[i + 1] + 1, &ne, &nesc);
                if (g.key_cmp(m.key_b, m.key_e, m.key_esc, keypos[i + 1] + 1, ne, nesc) == 0) continue;
            }
            uint32_t st = gelf_dispatch<EMIT>(g, m, r, t, slot, &cnt);
            if (st != G_OK) {
                r.status = st;
                return;
            }
        }
    } else {
        // selection rounds: each round re-walks the object and picks the smallest key greater
        // than the previous winner (among equal keys: the last one).
        bool have_prev = false;
        Member prev;
        for (;;) {
            bool have_best = false;
            Member best;
            uint32_t p = g.skip_ws(0) + 1;
            for (uint32_t k = 0; k < n; ++k) {
                p = g.skip_ws(p);
                if (k) p = g.skip_ws(p + 1);  // the ','
                Member m;
                g.member(p, &m);
                p = m.end;
                if (have_prev && g.key_cmp(m.key_b, m.key_e, m.key_esc, prev.key_b, prev.key_e, prev.key_esc) <= 0) continue;
                if (!have_best || g.key_cmp(m.key_b, m.key_e, m.key_esc, best.key_b, best.key_e, best.key_esc) <= 0) {
                    best = m;
                    have_best = true;
                }
            }
            if (!have_best) break;
            uint32_t st = gelf_dispatch<EMIT>(g, best, r, t, slot, &cnt);
            if (st != G_OK) {
                r.status = st;
                return;
            }
            prev = best;
            have_prev = true;
        }
    }
    if (!EMIT) r.n_ent = cnt;
}

template <class R>
__device__ void gelf_line(R& rd, uint32_t len, uint32_t* keypos, uint32_t* stack, GRow& r, const DevTables& t) {
    Gelf<R> g{rd, len, false, stack};
    bool is_object = false;
    uint32_t e = gelf_validate(g, keypos, r, &is_object);
    if (e == J_CTRL) {  // gelf_decoder.rs:44-46
        g.retry = true;
        r = GRow();
        e = gelf_validate(g, keypos, r, &is_object);
        if (e == J_OK) r.flags |= FG_F_GELF_RETRY;
    }
    if (e != J_OK) {
        r.status = G_JSON;
        return;
    }
    if (!is_object) {
        r.status = G_EMPTY;
        return;
    }
    gelf_sorted_dispatch<false>(g, keypos, r, t, 0);
    if (r.status != G_OK) return;
    if (!r.have_ts) r.flags |= FG_F_TS_NOW;  // :109
    if (!r.have_host) r.status = G_NOHOST;   // :110
}

// =============================================================================================
// The fast form
// =============================================================================================
constexpr uint32_t kGelfLines = 32;       // lines per group (cap): the member arrays below are per lane
constexpr uint32_t kFastMembers = 16;
// per lane: 16 sort keys (u64) | 16 records (u64) | 16 order bytes  = 272 bytes; the general
// form's keypos[32] + nesting stack alias the first 192 bytes of the same block
constexpr uint32_t kLaneBlock = kFastMembers * 16u + 16u;
constexpr uint32_t kGelfExtraLds = kGelfLines * kLaneBlock;
static_assert(kMaxStored * 4u + kMaxDepth / 8u <= kLaneBlock, "general-form arrays must fit the lane block");

enum : uint32_t { K_TS = 0, K_HOST = 1, K_SHORT = 2, K_FULL = 3, K_VERSION = 4, K_LEVEL = 5, K_OTHER = 6 };

// record: key_b | key_len << 16 | v_b << 32 | kind << 48 | v_esc << 52 ; strings: v_len in the
// stash word, numbers / bool: the value bits in the stash word
__device__ __forceinline__ uint64_t rec_pack(uint32_t key_b, uint32_t key_len, uint32_t v_b, uint32_t kind, uint32_t v_esc) {
    return (uint64_t)key_b | ((uint64_t)key_len << 16) | ((uint64_t)v_b << 32) | ((uint64_t)kind << 48) | ((uint64_t)v_esc << 52);
}

struct GelfFormat {
    static constexpr bool kStageABitmap = true;
    uint8_t* lane_blocks;  // LDS: kGelfLines x kLaneBlock

    // stage A: '"' | '\\' | control characters
    static __device__ __forceinline__ uint32_t mask16(const uint4& v) {
        const uint32_t Q = 0x22222222u, B = 0x5C5C5C5Cu;
        return gather16(eq_flags(v.x, Q) | eq_flags(v.x, B) | ctrl_flags(v.x), eq_flags(v.y, Q) | eq_flags(v.y, B) | ctrl_flags(v.y),
                        eq_flags(v.z, Q) | eq_flags(v.z, B) | ctrl_flags(v.z), eq_flags(v.w, Q) | eq_flags(v.w, B) | ctrl_flags(v.w));
    }

    // A 16-byte register window consumed byte by byte.
    struct Win {
        uint64_t lo, hi;
        uint32_t left;  // bytes still in view
        __device__ __forceinline__ uint32_t peek() const { return (uint32_t)lo & 0xFFu; }
        __device__ __forceinline__ void pop() {
            lo = (lo >> 8) | (hi << 56);
            hi >>= 8;
            --left;
        }
    };
    static __device__ __forceinline__ Win window(const Tile& T, uint32_t base, uint32_t p, uint32_t len) {
        uint32_t w[4];
        load16(T, base + p, w);
        Win x;
        x.lo = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
        x.hi = (uint64_t)w[2] | ((uint64_t)w[3] << 32);
        const uint32_t avail = p < len ? len - p : 0u;
        x.left = avail < 16u ? avail : 16u;
        return x;
    }
    static __device__ __forceinline__ bool ws(uint32_t c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n'; }
    // skip whitespace inside the window; false = the view ran out (caller leaves the fast form)
    static __device__ __forceinline__ bool skip(Win& x, uint32_t& p) {
        while (x.left && ws(x.peek())) {
            x.pop();
            ++p;
        }
        return x.left != 0;
    }
    static __device__ __forceinline__ bool hex4_ok(uint32_t v) {  // four ASCII hex digits in a dword
        // per byte: '0'..'9' or ((c|0x20) in 'a'..'f')
        const uint32_t d = v ^ 0x30303030u;                                   // digits -> 0..9
        const uint32_t dig_bad = (d | ((d & 0x7F7F7F7Fu) + 0x76767676u)) & 0x80808080u;
        const uint32_t l = (v | 0x20202020u) ^ 0x60606060u;                   // a..f -> 1..6
        const uint32_t let_hi = (l | ((l & 0x7F7F7F7Fu) + 0x79797979u)) & 0x80808080u;  // > 6
        const uint32_t let_zero = ~(((l & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | l) & 0x80808080u;  // == 0
        return ((dig_bad & (let_hi | let_zero)) == 0u);
    }

    // String body from p (just past the opening quote): *end = index of the closing quote.
    // false = leave the fast form (raw control character, bad / unusual escape, end of line).
    static __device__ __forceinline__ bool skip_string(const Tile& T, LdsReader& rd, uint32_t base, uint32_t p, uint32_t len,
                                                       uint32_t* end, uint32_t* esc) {
        uint32_t has_esc = 0;
        for (;;) {
            const uint32_t h = find_bit_long(T.bm, base, p, len);
            if (h >= len) return false;
            uint32_t b0, b1;
            load8(T, base + h, &b0, &b1);
            const uint32_t c = b0 & 0xFFu;
            if (c == '"') {
                *end = h;
                *esc = has_esc;
                return true;
            }
            if (c != '\\') return false;  // raw control character: the general form handles the retry
            has_esc = 1;
            const uint32_t e = (b0 >> 8) & 0xFFu;
            if (h + 1u >= len) return false;
            if (e == '"' || e == '\\' || e == '/' || e == 'b' || e == 'f' || e == 'n' || e == 'r' || e == 't') {
                p = h + 2u;
                continue;
            }
            if (e != 'u' || h + 6u > len) return false;
            // \uXXXX: bytes 2..5 of the view
            const uint32_t hx = (b0 >> 16) | (b1 << 16);
            if (!hex4_ok(hx)) return false;
            const uint32_t d0 = hx & 0xFFu, d1 = (hx >> 8) & 0xFFu;
            const bool is_d = (d0 | 0x20u) == 'd';
            const uint32_t d1l = d1 | 0x20u;
            const bool high = is_d && (d1 == '8' || d1 == '9' || d1l == 'a' || d1l == 'b');
            const bool low = is_d && (d1l >= 'c' && d1l <= 'f');
            if (low) return false;  // lone low surrogate
            if (high) {             // must be followed by \uDC00..\uDFFF
                if (h + 12u > len) return false;
                uint32_t c0, c1;
                load8(T, base + h + 6u, &c0, &c1);
                if ((c0 & 0xFFFFu) != (('u' << 8) | '\\')) return false;
                const uint32_t hx2 = (c0 >> 16) | (c1 << 16);
                if (!hex4_ok(hx2)) return false;
                const uint32_t e0 = hx2 & 0xFFu, e1 = ((hx2 >> 8) & 0xFFu) | 0x20u;
                if (!((e0 | 0x20u) == 'd' && e1 >= 'c' && e1 <= 'f')) return false;
                p = h + 12u;
            } else {
                p = h + 6u;
            }
        }
    }

    static __device__ __forceinline__ uint32_t known_key(uint32_t n, const uint32_t w[4]) {
        // keys compared as (length, little-endian dwords of the first 16 bytes)
        if (n == 9u && w[0] == 0x656D6974u && w[1] == 0x6D617473u && (w[2] & 0xFFu) == 'p') return K_TS;                  // timestamp
        if (n == 4u && w[0] == 0x74736F68u) return K_HOST;                                                                // host
        if (n == 13u && w[0] == 0x726F6873u && w[1] == 0x656D5F74u && w[2] == 0x67617373u && (w[3] & 0xFFu) == 'e') return K_SHORT;  // short_message
        if (n == 12u && w[0] == 0x6C6C7566u && w[1] == 0x73656D5Fu && w[2] == 0x65676173u) return K_FULL;                 // full_message
        if (n == 7u && w[0] == 0x73726576u && (w[1] & 0xFFFFFFu) == 0x6E6F69u) return K_VERSION;                          // version
        if (n == 5u && w[0] == 0x6576656Cu && (w[1] & 0xFFu) == 'l') return K_LEVEL;                                      // level
        return K_OTHER;
    }

    // serde_json 0.8 number scanning (fg_numparse.hpp json_number) for the everyday shape -- optional
    // '-', 1..18 digits with at most one '.', no exponent -- done on registers, 16 bytes per LDS
    // round trip.  Returns true = parsed (*kind / *bits as json_number would give, *end = index past
    // the token), false = not this shape (the caller runs json_number, which owns every error).
    static __device__ __forceinline__ bool fast_decimal(const Tile& T, uint32_t base, uint32_t p, uint32_t len, uint32_t* kind,
                                                        uint64_t* bits, uint32_t* end) {
        Win x = window(T, base, p, len);
        bool neg = false;
        if (x.left && x.peek() == '-') {
            neg = true;
            x.pop();
            ++p;
        }
        uint64_t sig = 0;
        uint32_t nd = 0, nf = 0, first = 0x100u, c = 0x100u;
        bool dot = false;
        for (;;) {
            if (x.left == 0) {
                if (p >= len) break;  // (c stays "end of line")
                x = window(T, base, p, len);
            }
            c = x.peek();
            const uint32_t d = c - '0';
            if (d <= 9u) {
                if (nd == 0) first = d;
                sig = sig * 10u + d;
                ++nd;
                nf += dot ? 1u : 0u;
            } else if (c == '.' && !dot && nd != 0) {
                dot = true;
            } else {
                break;
            }
            x.pop();
            ++p;
            c = 0x100u;
        }
        if (c == 'e' || c == 'E' || c == '.' || nd == 0 || nd > 18u) return false;
        if (dot && nf == 0) return false;                 // "1." is an error: json_number reports it
        if (first == 0u && (nd - nf) > 1u) return false;  // leading zero rule: ditto
        *end = p;
        if (dot) {
            *kind = V_F64;
            return num::json_f64_from_parts(!neg, sig, -(int32_t)nf, bits);
        }
        if (!neg || sig == 0) {
            *kind = V_U64;  // "-0" -> visit_i64(0) -> U64(0)
            *bits = sig;
        } else {
            *kind = V_I64;
            *bits = 0ull - sig;
        }
        return true;
    }

    // 8 bytes at line index p as a u64 + the mask (8 bits) of the bytes that are NOT a space; bytes
    // past the end of the line read as spaces.  Only ' ' counts as inter-token whitespace on the fast
    // form: TAB / CR / LF between tokens fail the "expected character" tests and leave it.
    static __device__ __forceinline__ uint64_t punct8(const Tile& T, uint32_t base, uint32_t p, uint32_t len, uint32_t* nonspace) {
        uint32_t lo, hi;
        load8(T, base + p, &lo, &hi);
        const uint32_t sp = (__builtin_amdgcn_udot4(eq_flags(lo, 0x20202020u), 0x08040201u, 0u, false) +
                             __builtin_amdgcn_udot4(eq_flags(hi, 0x20202020u), 0x80402010u, 0u, false)) >> 7;
        const uint32_t avail = p < len ? len - p : 0u;
        const uint32_t inside = avail >= 8u ? 0xFFu : (1u << avail) - 1u;
        *nonspace = ~sp & inside;
        return (uint64_t)lo | ((uint64_t)hi << 32);
    }
    static __device__ __forceinline__ uint32_t byte_of(uint64_t v, uint32_t i) { return (uint32_t)(v >> (8u * i)) & 0xFFu; }

    // Walk the line; members -> keys[] / recs[] (LDS, this lane) + the value word in the stash.
    // Returns the member count, or 0xFFFFFFFF = not the fast form.  Mostly straight-line: per member
    // one bit scan for the key's closing quote, one 8-byte register window for `[sp]:[sp]<value>`,
    // the value (bit scan / register decimal / literal compare), one window for `[sp],[sp]"` | `[sp]}`.
    __device__ __forceinline__ uint32_t fast_walk(const Tile& T, uint32_t base, uint32_t len, uint64_t* keys, uint64_t* recs,
                                                  uint64_t* stash) const {
        constexpr uint32_t BAIL = 0xFFFFFFFFu;
        LdsReader rd(T.w, base);
        uint32_t ns;
        uint64_t w = punct8(T, base, 0, len, &ns);
        if (ns == 0) return BAIL;
        uint32_t i = (uint32_t)__builtin_ctz(ns);
        if (byte_of(w, i) != '{') return BAIL;
        ns &= ns - 1u;
        uint32_t p;        // index of the next key's opening quote
        bool closed = false;
        uint32_t after = 0;  // index just past the closing '}'
        if (ns == 0) {
            // "{" then 7+ spaces, or a short line: look again from there
            w = punct8(T, base, i + 1u, len, &ns);
            if (ns == 0) return BAIL;
            const uint32_t j = (uint32_t)__builtin_ctz(ns);
            const uint32_t c = byte_of(w, j);
            if (c == '}') {
                closed = true;
                after = i + 1u + j + 1u;
            } else if (c != '"') {
                return BAIL;
            }
            p = i + 1u + j;
        } else {
            const uint32_t j = (uint32_t)__builtin_ctz(ns);
            const uint32_t c = byte_of(w, j);
            if (c == '}') {
                closed = true;
                after = j + 1u;
            } else if (c != '"') {
                return BAIL;
            }
            p = j;
        }
        // One member per iteration.  Written as predicated straight-line code with a sticky `ok`
        // instead of early returns: on this hardware every divergent exit costs a dozen scalar
        // instructions of exec-mask bookkeeping, and there would be twenty of them per member.
        uint32_t n = 0;
        bool ok = true;
        while (ok && !closed) {
            ok = n < kFastMembers;
            // ---- key: no escapes, no control characters ------------------------------------------
            const uint32_t key_b = p + 1u;
            const uint32_t key_e = find_bit(T.bm, base, key_b, len);
            ok = ok && key_e < len;
            // ---- '"' [sp] ':' [sp] value-start, all inside 8 bytes ---------------------------------
            w = punct8(T, base, key_e, len, &ns);
            ok = ok && (uint32_t)(w & 0xFFu) == '"';  // else the hit was a '\\' or a control character
            ns &= ~1u;                               // the quote itself
            const uint32_t i1 = ns ? (uint32_t)__builtin_ctz(ns) : 0u;
            ok = ok && ns != 0u && byte_of(w, i1) == ':';
            ns &= ns - 1u;
            const uint32_t i2 = ns ? (uint32_t)__builtin_ctz(ns) : 0u;
            ok = ok && ns != 0u;
            const uint32_t c = byte_of(w, i2);
            const uint32_t v = key_e + i2;
            uint32_t kind = V_NULL, v_b = v, v_esc = 0, vend = v;
            uint64_t word = 0;
            const bool is_str = c == '"', is_num = c == '-' || (c - '0') <= 9u;
            if (ok && is_str) {
                kind = V_STRING;
                v_b = v + 1u;
                uint32_t e = 0;
                ok = skip_string(T, rd, base, v_b, len, &e, &v_esc);
                word = e - v_b;
                vend = e + 1u;
            }
            if (ok && is_num) {
                uint32_t k2 = 0;
                bool good = fast_decimal(T, base, v, len, &k2, &word, &vend);
                if (!good) good = num::json_number(rd, v, len, &vend, &k2, &word);
                ok = good;
                kind = k2;
            }
            if (!is_str && !is_num) {
                uint32_t l0, l1;
                load8(T, base + v, &l0, &l1);
                const bool t = c == 't' && l0 == 0x65757274u && v + 4u <= len;                           // "true"
                const bool f = c == 'f' && l0 == 0x736C6166u && (l1 & 0xFFu) == 'e' && v + 5u <= len;    // "false"
                const bool nl = c == 'n' && l0 == 0x6C6C756Eu && v + 4u <= len;                          // "null"
                ok = ok && (t || f || nl);  // else: nested value or garbage
                kind = nl ? V_NULL : V_BOOL;
                word = t ? 1u : 0u;
                vend = v + (f ? 5u : 4u);
            }
            // ---- record + sort key ---------------------------------------------------------------
            if (ok) {
                uint32_t k0, k1;
                load8(T, base + key_b, &k0, &k1);
                const uint32_t kl = key_e - key_b;
                uint64_t pre = (uint64_t)k0 | ((uint64_t)k1 << 32);
                if (kl < 8u) pre &= kl == 0u ? 0ull : (~0ull >> (64u - 8u * kl));
                // big-endian first 7 bytes in bits 63..8, member index in bits 7..0
                const uint64_t be = __builtin_bswap64(pre);
                keys[n] = (be & ~0xFFull) | n;
                recs[n] = rec_pack(key_b, kl, v_b, kind, v_esc);
                stash[n * kWave + threadIdx.x] = word;
            }
            ++n;
            // ---- [sp] ',' [sp] '"'   |   [sp] '}' ---------------------------------------------------
            w = punct8(T, base, vend, len, &ns);
            const uint32_t j1 = ns ? (uint32_t)__builtin_ctz(ns) : 0u;
            const uint32_t d = byte_of(w, j1);
            ok = ok && ns != 0u && (d == '}' || d == ',');
            closed = d == '}';
            after = vend + j1 + 1u;
            ns &= ns - 1u;
            const uint32_t j2 = ns ? (uint32_t)__builtin_ctz(ns) : 0u;
            ok = ok && (closed || (ns != 0u && byte_of(w, j2) == '"'));
            p = vend + j2;
        }
        if (!ok) return BAIL;
        // trailing spaces only (anything else, incl. other whitespace: general form decides)
        while (after < len) {
            w = punct8(T, base, after, len, &ns);
            if (ns != 0) return BAIL;
            after += 8u;
        }
        return n;
    }

    __device__ __forceinline__ RowOut decode(const GroupCtx& c, const DevTables& t) const {
        const uint32_t lane = threadIdx.x;
        const uint32_t len = (uint32_t)(c.o1 - c.o0);
        const bool in_tile = (c.o1 - c.a0) <= (uint64_t)c.span;
        const uint32_t base = (uint32_t)(c.o0 - c.a0);
        Tile T{reinterpret_cast<const uint32_t*>(c.smem), reinterpret_cast<const uint32_t*>(c.bm16)};
        uint8_t* blk = lane_blocks + (lane < kGelfLines ? lane : 0u) * kLaneBlock;
        uint64_t* keys = reinterpret_cast<uint64_t*>(blk);
        uint64_t* recs = keys + kFastMembers;
        uint8_t* order = blk + kFastMembers * 16u;
        uint32_t* keypos = reinterpret_cast<uint32_t*>(blk);              // general form (aliases keys)
        uint32_t* stack = reinterpret_cast<uint32_t*>(blk + kMaxStored * 4u);

        GRow r;
        uint32_t nm = 0xFFFFFFFFu;  // members found by the fast form
        if (c.valid && in_tile && len < 65536u && c.stash) nm = fast_walk(T, base, len, keys, recs, c.stash);
        bool fast = c.valid && nm != 0xFFFFFFFFu;
        if (c.ablate & 16u) {  // measurement only: the walk alone
            RowOut z{};
            z.meta = nm;
            return z;
        }
        uint32_t sorted_n = 0;
        if (c.ablate & 32u) {  // measurement only: (same point as 16 since numbers are converted in the walk)
            RowOut z{};
            z.meta = nm + (fast ? 1u : 0u);
            return z;
        }
        if (fast) {
            // ---- rank the keys in registers (BTreeMap order; equal keys: later member last) ------
            uint64_t k[kFastMembers];
#pragma unroll
            for (uint32_t j = 0; j < kFastMembers; ++j) k[j] = j < nm ? keys[j] : ~0ull;
            for (uint32_t i = 0; i < nm; ++i) {  // (k[] stays in registers: only the inner loop is unrolled)
                const uint64_t ki = keys[i];
                uint32_t rank = 0;
#pragma unroll
                for (uint32_t j = 0; j < kFastMembers; ++j) rank += k[j] < ki ? 1u : 0u;
                order[rank] = (uint8_t)i;
            }
            // ---- duplicates / unresolved order: adjacent keys with the same 7-byte prefix ---------
            LdsReader rd(T.w, base);
            for (uint32_t s = 0; s + 1u < nm && fast; ++s) {
                const uint32_t i = order[s], j = order[s + 1u];
                if ((keys[i] >> 8) != (keys[j] >> 8)) continue;
                const uint64_t ri = recs[i], rj = recs[j];
                const uint32_t li = (uint32_t)(ri >> 16) & 0xFFFFu, lj = (uint32_t)(rj >> 16) & 0xFFFFu;
                bool same = li == lj;
                for (uint32_t q = 7; q < li && same; ++q) same = rd.byte(((uint32_t)ri & 0xFFFFu) + q) == rd.byte(((uint32_t)rj & 0xFFFFu) + q);
                if (!same) fast = false;  // two different keys share 7 bytes: the general form orders them
                else order[s] = 0xFFu;    // earlier duplicate: skipped (the last one wins, BTreeMap::insert)
            }
            sorted_n = nm;
        }
        if (c.ablate & 64u) {  // measurement only: walk + numbers + ranking
            RowOut z{};
            z.meta = nm + sorted_n + (fast ? 1u : 0u) + order[0];
            return z;
        }
        if (fast) {
            // ---- gelf_decoder.rs:51-106 in sorted key order -------------------------------------------
            uint32_t cnt = 0;
            for (uint32_t s = 0; s < sorted_n; ++s) {
                const uint32_t i = order[s];
                if (i == 0xFFu) continue;
                const uint64_t rec = recs[i];
                const uint32_t key_b = (uint32_t)rec & 0xFFFFu, kl = (uint32_t)(rec >> 16) & 0xFFFFu;
                const uint32_t v_b = (uint32_t)(rec >> 32) & 0xFFFFu, kind = (uint32_t)(rec >> 48) & 0xFu, v_esc = (uint32_t)(rec >> 52) & 1u;
                uint32_t w[4];
                load16(T, base + key_b, w);
                const uint32_t which = known_key(kl, w);
                if (which == K_OTHER) {
                    ++cnt;  // (nested values never reach the fast form)
                    continue;
                }
                const uint64_t word = c.stash[i * kWave + lane];
                uint32_t st = G_OK;
                if (which == K_TS) {
                    if (kind == V_F64) r.ts = num::bits_to_f64(word);
                    else if (kind == V_U64) r.ts = (double)word;
                    else if (kind == V_I64) r.ts = (double)(int64_t)word;
                    else st = G_TS;
                    r.have_ts = 1;
                } else if (which == K_HOST) {
                    if (kind != V_STRING) st = G_HOST;
                    r.host_off = v_b;
                    r.host_len = (uint32_t)word;
                    r.have_host = 1;
                    if (v_esc) r.flags |= FG_F_HOST_ESC;
                } else if (which == K_SHORT) {
                    if (kind != V_STRING) st = G_SHORT;
                    r.msg_off = v_b;
                    r.msg_len = (uint32_t)word;
                    if (v_esc) r.flags |= FG_F_MSG_ESC;
                } else if (which == K_FULL) {
                    if (kind != V_STRING) st = G_FULL;
                    r.full_off = v_b;
                    r.full_len = (uint32_t)word;
                    if (v_esc) r.flags |= FG_F_FULLMSG_ESC;
                } else if (which == K_VERSION) {
                    if (kind != V_STRING) st = G_VERSTR;
                    else {
                        // "1.0" / "1.1" -- by DECODED value: an escaped spelling goes through the general form
                        if (v_esc) {
                            fast = false;
                            break;
                        }
                        uint32_t v0, v1;
                        load8(T, base + v_b, &v0, &v1);
                        const uint32_t three = v0 & 0xFFFFFFu;
                        if (!((uint32_t)word == 3u && (three == 0x302E31u || three == 0x312E31u))) st = G_VER;
                    }
                } else {  // level
                    if (kind != V_U64) st = G_LEVEL;  // Value::as_u64 (NumCast): floats and negatives -> None
                    else if (word > 7) st = G_LEVEL7;
                    else r.severity = (uint32_t)word;
                }
                if (st != G_OK) {
                    r.status = st;
                    break;
                }
            }
            if (fast && r.status == G_OK) {
                if (!r.have_ts) r.flags |= FG_F_TS_NOW;  // :109
                if (!r.have_host) r.status = G_NOHOST;   // :110
                r.n_ent = r.status == G_OK ? cnt : 0u;
            }
        }
        const bool general = c.valid && !fast;
        if (general) {  // the exact general form (rare)
            r = GRow();
            if (in_tile) {
                LdsReader rd(T.w, base);
                gelf_line(rd, len, keypos, stack, r, t);
            } else {
                GlobalReader rd(reinterpret_cast<const uint32_t*>(c.bytes), c.o0);
                gelf_line(rd, len, keypos, stack, r, t);
            }
            if (r.status != G_OK) r.n_ent = 0;
        }
        bool overflow;
        const uint32_t first = alloc_entries(t, r.n_ent, &overflow);
        if (overflow) {
            r.status = FG_ST_OVERFLOW;
            r.n_ent = 0;
        }
        if (r.n_ent != 0) {
            if (!general) {
                uint32_t k = first;
                for (uint32_t s = 0; s < sorted_n; ++s) {
                    const uint32_t i = order[s];
                    if (i == 0xFFu) continue;
                    const uint64_t rec = recs[i];
                    const uint32_t key_b = (uint32_t)rec & 0xFFFFu, kl = (uint32_t)(rec >> 16) & 0xFFFFu;
                    uint32_t w[4];
                    load16(T, base + key_b, w);
                    if (known_key(kl, w) != K_OTHER) continue;
                    const uint32_t v_b = (uint32_t)(rec >> 32) & 0xFFFFu, kind = (uint32_t)(rec >> 48) & 0xFu, v_esc = (uint32_t)(rec >> 52) & 1u;
                    const uint64_t word = c.stash[i * kWave + lane];
                    t.ent_name[k] = fg_span{key_b, kl};
                    t.ent_type[k] = (uint8_t)kind;
                    t.ent_val[k] = kind == V_STRING ? ((uint64_t)v_b | (word << 32)) : kind == V_NULL ? 0ull : word;
                    t.ent_flags[k] = (uint8_t)((kind == V_STRING && v_esc) ? FG_EF_VAL_ESC : 0);
                    ++k;
                }
            } else {
                GRow tmp = r;
                if (in_tile) {
                    LdsReader rd(T.w, base);
                    Gelf<LdsReader> g{rd, len, (r.flags & FG_F_GELF_RETRY) != 0, stack};
                    gelf_sorted_dispatch<true>(g, keypos, tmp, t, first);
                } else {
                    GlobalReader rd(reinterpret_cast<const uint32_t*>(c.bytes), c.o0);
                    Gelf<GlobalReader> g{rd, len, (r.flags & FG_F_GELF_RETRY) != 0, stack};
                    gelf_sorted_dispatch<true>(g, keypos, tmp, t, first);
                }
            }
        }
        RowOut o;
        const bool ok = r.status == G_OK;
        const fg_span none{0, FG_NONE};
        o.meta = r.status | (0xFFu << 8) | ((ok ? r.severity : 0xFFu) << 16) | ((ok ? r.flags : 0u) << 24);
        o.ts = (ok && r.have_ts) ? r.ts : 0.0;
        o.span[S_HOST] = ok ? fg_span{r.host_off, r.host_len} : none;
        o.span[S_APP] = none;
        o.span[S_PROC] = none;
        o.span[S_MSGID] = none;
        o.span[S_MSG] = ok ? fg_span{r.msg_off, r.msg_len} : none;
        o.span[S_FULL] = ok ? fg_span{r.full_off, r.full_len} : none;
        o.first = first;
        o.count = r.n_ent;
        return o;
    }
};

template <int NB, bool PROF>
__global__ __launch_bounds__(kWave, 2) void k_gelf(const uint8_t* __restrict__ bytes, const uint64_t* __restrict__ offsets,
                                                  uint64_t n, DevTables t, uint32_t tile_cap, uint32_t L, uint64_t groups,
                                                  unsigned long long* prof, uint64_t* stash_base, FrameArgs fr) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    GelfFormat fmt{smem + tile_cap + 64u + (tile_cap / 16u + 16u) * 2u};
    persistent_loop<NB, PROF>(bytes, offsets, n, t, tile_cap, L, groups, prof, stash_base, fmt, fr);
}

}  // namespace fg

extern "C" int fg_launch_gelf(const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t n, const fg::DevTables* t,
                              uint64_t avg_len, hipStream_t stream, uint64_t* stash, uint32_t stash_blocks, uint32_t strip,
                              const uint8_t* line_bad) {
    if (n == 0) return 0;
    fg::LaunchPlan p;
    // at most kGelfLines lines per group: plan with twice the average length (L <= 32 follows), then
    // size the tile for the real one
    if (fg::plan_launch(fg::k_gelf<fg::kComputeBoundWindow, false>, n, avg_len, fg::kGelfExtraLds, 57344u, stash ? stash_blocks : 0u, &p, fg::kGelfLines))
        return -1;
    if (stash_blocks == 0) stash = nullptr;
    dim3 grid(p.blocks), block(fg::kWave);
    if (getenv("FG_PROF")) {
        fg::ProfRun pr;
        if (!pr.begin(stream)) return -1;
        hipLaunchKernelGGL((fg::k_gelf<fg::kComputeBoundWindow, true>), grid, block, p.lds, stream, d_bytes, d_offsets, n, *t, p.tile, p.L,
                           p.groups, pr.d, stash, fg::FrameArgs{strip, line_bad});
        pr.end(stream, "gelf", p);
        return (int)hipGetLastError();
    }
    hipLaunchKernelGGL((fg::k_gelf<fg::kComputeBoundWindow, false>), grid, block, p.lds, stream, d_bytes, d_offsets, n, *t, p.tile, p.L,
                       p.groups, (unsigned long long*)nullptr, stash, fg::FrameArgs{strip, line_bad});
    return (int)hipGetLastError();
}
