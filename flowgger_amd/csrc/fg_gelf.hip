// fg_gelf.hip -- gfx950 kernel for GelfDecoder::decode
// (reference: src/flowgger/decoder/gelf_decoder.rs:34-125; JSON semantics = serde_json 0.8).
//
// One wave per 64 lines; the group's bytes are streamed HBM -> LDS (coalesced 16 B/lane) and each
// lane parses ITS line out of LDS:
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
#include "fg_device.hpp"
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

__global__ __launch_bounds__(kWave) void k_gelf(const uint8_t* __restrict__ bytes, const uint64_t* __restrict__ offsets,
                                               uint64_t n, DevTables t, uint32_t tile_cap) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t lane = threadIdx.x;
    uint32_t* keypos = reinterpret_cast<uint32_t*>(smem + tile_cap + 64u) + lane * kMaxStored;
    uint32_t* stack = reinterpret_cast<uint32_t*>(smem + tile_cap + 64u + kWave * kMaxStored * 4u) + lane * (kMaxDepth / 32u);
    const uint64_t l0 = (uint64_t)blockIdx.x * kWave;
    const uint64_t li = l0 + lane;
    const bool valid = li < n;
    const uint64_t last = (l0 + kWave < n) ? l0 + kWave : n;
    const uint64_t o0 = offsets[valid ? li : last];
    const uint64_t o1 = offsets[valid ? li + 1 : last];
    const uint64_t lo = __shfl(o0, 0, kWave);
    const uint64_t hi = __shfl(o1, (int)(last - l0 - 1), kWave);
    const uint64_t a0 = lo & ~15ull;
    const uint64_t want = hi - a0;
    const uint32_t span = want > tile_cap ? tile_cap : (uint32_t)((want + 15ull) & ~15ull);
    stage_tile(bytes, a0, span, smem);
    __syncthreads();

    GRow r;
    const uint32_t len = (uint32_t)(o1 - o0);
    const bool in_tile = (o1 - a0) <= (uint64_t)span;
    const uint32_t base = (uint32_t)(o0 - a0);
    if (valid) {
        if (in_tile) {
            LdsReader rd(reinterpret_cast<const uint32_t*>(smem), base);
            gelf_line(rd, len, keypos, stack, r, t);
        } else {
            GlobalReader rd(reinterpret_cast<const uint32_t*>(bytes), o0);
            gelf_line(rd, len, keypos, stack, r, t);
        }
        if (r.status != G_OK) r.n_ent = 0;
    }
    uint32_t total;
    uint32_t ex = wave_exclusive_sum(r.n_ent, &total);
    uint32_t first = 0;
    if (total != 0) {
        unsigned long long slot0 = 0;
        if (lane == 0) slot0 = atomicAdd(t.ent_used, (unsigned long long)total);
        slot0 = __shfl(slot0, 0, kWave);
        unsigned long long mine = slot0 + ex;
        if (r.n_ent != 0) {
            if (mine + r.n_ent > t.ent_cap) {
                r.status = FG_ST_OVERFLOW;
                r.n_ent = 0;
            } else {
                first = (uint32_t)mine;
                GRow tmp = r;
                if (in_tile) {
                    LdsReader rd(reinterpret_cast<const uint32_t*>(smem), base);
                    Gelf<LdsReader> g{rd, len, (r.flags & FG_F_GELF_RETRY) != 0, stack};
                    gelf_sorted_dispatch<true>(g, keypos, tmp, t, first);
                } else {
                    GlobalReader rd(reinterpret_cast<const uint32_t*>(bytes), o0);
                    Gelf<GlobalReader> g{rd, len, (r.flags & FG_F_GELF_RETRY) != 0, stack};
                    gelf_sorted_dispatch<true>(g, keypos, tmp, t, first);
                }
            }
        }
    }
    if (valid) {
        const bool ok = r.status == G_OK;
        const fg_span none{0, FG_NONE};
        t.meta[li] = r.status | (0xFFu << 8) | ((ok ? r.severity : 0xFFu) << 16) | ((ok ? r.flags : 0u) << 24);
        t.ts[li] = (ok && r.have_ts) ? r.ts : 0.0;
        t.span[S_HOST][li] = ok ? fg_span{r.host_off, r.host_len} : none;
        t.span[S_APP][li] = none;
        t.span[S_PROC][li] = none;
        t.span[S_MSGID][li] = none;
        t.span[S_MSG][li] = ok ? fg_span{r.msg_off, r.msg_len} : none;
        t.span[S_FULL][li] = ok ? fg_span{r.full_off, r.full_len} : none;
        t.ent_first[li] = first;
        t.ent_count[li] = r.n_ent;
    }
}

}  // namespace fg

extern "C" int fg_launch_gelf(const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t n, const fg::DevTables* t,
                              uint32_t tile_cap, hipStream_t stream) {
    if (n == 0) return 0;
    uint64_t groups = (n + fg::kWave - 1) / fg::kWave;
    if (groups > 0x7FFFFFFFull) return -1;
    uint32_t lds = tile_cap + 64u + fg::kWave * fg::kMaxStored * 4u + fg::kWave * (fg::kMaxDepth / 8u);
    hipLaunchKernelGGL(fg::k_gelf, dim3((uint32_t)groups), dim3(fg::kWave), lds, stream, d_bytes, d_offsets, n, *t, tile_cap);
    return (int)hipGetLastError();
}
