// fg_gelf.hip -- gfx950 kernel for GelfDecoder::decode
// (reference: src/flowgger/decoder/gelf_decoder.rs:34-125; JSON semantics = serde_json 0.8).
//
// Runs on the streaming pipeline of fg_pipeline.hpp (persistent waves, register prefetch window, LDS tile per line group).
//
// FAST FORM = fg_gelf2.hpp, wave-cooperative: stage A writes five class bitmaps of the tile, a word pass derives the real
// quotes / the in-string mask / the structural characters outside strings with wave-wide scans, and then the work item is
// ONE OBJECT MEMBER per lane (not one line per lane): delimiting, number parsing, BTreeMap ranking, the dispatch of
// gelf_decoder.rs:51-106 and the entry stores all run with 64 busy lanes whatever the members per line.  It handles flat
// objects without escaped keys -- every GELF producer's output -- and hands anything else back untouched.
//
// GENERAL FORM (below; lane-per-line, exact, rare): the three-pass byte-walking implementation --
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
#include "fg_fused.hpp"
#include "fg_numparse.hpp"
#include "fg_gelf2.hpp"

namespace fg {

using namespace gelf2;  // G_* status codes (== the reference's error strings), V_* value kinds

enum : uint32_t { J_OK = 0, J_SYNTAX = 1, J_CTRL = 2 };
constexpr uint32_t V_NESTED = 6;
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
__device__ __forceinline__ uint32_t gelf_dispatch(const Gelf<R>& g, const Member& m, GRow& r, const DevTables& t, uint32_t slot,
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
__device__ __forceinline__ uint32_t gelf_validate(const Gelf<R>& g, uint32_t* keypos, GRow& r, bool* is_object) {
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
__device__ __forceinline__ void gelf_sorted_dispatch(const Gelf<R>& g, uint32_t* keypos, GRow& r, const DevTables& t, uint32_t slot) {
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
__device__ __forceinline__ void gelf_line(R& rd, uint32_t len, uint32_t* keypos, uint32_t* stack, GRow& r, const DevTables& t) {
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
// Kernel 1: the fast form (fg_gelf2.hpp) on the streaming pipeline.  Lines it hands back get the status kPending.
// Kernel 2: the exact general form, lane per line, for the pending lines (rare).
// Two kernels because the general form (recursive-descent shaped, 250 registers, a stack in scratch memory) would set
// the register budget -- hence the waves per SIMD -- of the fast form, which is latency bound and lives on occupancy.
// =============================================================================================
constexpr uint32_t kGelfLines = 64;       // lines per group (cap)
constexpr uint32_t kPending = 0xFCu;      // internal status between the two kernels; never visible to callers

inline uint32_t gelf_extra_lds(uint32_t tile, uint32_t lines) { return gelf2::extra_bytes(tile, lines); }

struct GelfFormat {
    static constexpr uint32_t kClasses = gelf2::kClasses;
    static __device__ __forceinline__ void classify_store(const uint4& q, uint16_t* bm16, uint32_t chunk, uint32_t stride, uint32_t term4) {
        uint32_t m[gelf2::kClasses + 1];
        gelf2::classify(q.x, q.y, q.z, q.w, m, term4);
        gelf2::store_classes(m, bm16, chunk, stride);
    }
    uint8_t* extra;  // LDS behind the class bitmaps
    uint32_t tile_cap, lines;
    unsigned long long* pacc;  // measurement build: the wave's phase clocks (LDS; a hot global atomic per phase would BE the profile)

    __device__ __forceinline__ RowOut decode(const GroupCtx& c, const DevTables& t) const {
        const uint32_t lane = threadIdx.x;
        const uint32_t len = (uint32_t)(c.o1 - c.o0);
        const bool in_tile = (c.o1 - c.a0) <= (uint64_t)c.span;
        const uint32_t base = (uint32_t)(c.o0 - c.a0);
        gelf2::Lds L = gelf2::carve(c.smem, c.bm16, tile_cap, extra, lines);
        L.ent_state = c.ent_state;
        L.alloc_chunk = t.alloc_chunk;
        const bool tile_lane = c.valid && in_tile && lane < lines && !(c.ablate & 4u);
        const gelf2::LineOut f = c.phase ? gelf2::decode_tile<true>(L, c.span, tile_lane, base, len, t, pacc)
                                         : gelf2::decode_tile<false>(L, c.span, tile_lane, base, len, t);
        RowOut o;
        const fg_span none{0, FG_NONE};
        const bool ok = f.handled && f.status == G_OK;
        o.meta = f.handled ? (f.status | (0xFFu << 8) | ((ok ? f.severity : 0xFFu) << 16) | ((ok ? f.flags : 0u) << 24)) : kPending;
        // (tell the general kernel that this launch left it something: any lane, the same value, idempotent)
        if (t.pending && c.valid && !f.handled) gstore(t.pending, 0, t.epoch);
        o.ts = (ok && f.have_ts) ? f.ts : 0.0;
        o.span[S_HOST] = ok ? fg_span{f.host_off, f.host_len} : none;
        o.span[S_APP] = none;
        o.span[S_PROC] = none;
        o.span[S_MSGID] = none;
        o.span[S_MSG] = ok ? fg_span{f.msg_off, f.msg_len} : none;
        o.span[S_FULL] = ok ? fg_span{f.full_off, f.full_len} : none;
        o.first = f.handled ? f.first : 0u;
        o.count = f.handled ? f.n_ent : 0u;
        return o;
    }
};

// TILE / LINES != 0: the geometry as compile-time constants (every LDS address becomes lane * k + immediate: without them the
// addresses of the class bitmaps alone were 12 hoisted registers, spilled to scratch and reloaded in front of every store)
template <int NB, bool PROF, int MINW = 4, uint32_t TILE = 0, uint32_t LINES = 0>
__global__ __launch_bounds__(kWave, MINW) void k_gelf(const uint8_t* __restrict__ bytes, const uint64_t* __restrict__ offsets,
                                                  uint64_t n, DevTables t, uint32_t tile_cap_, uint32_t L_, uint64_t groups,
                                                  unsigned long long* prof, uint64_t* stash_base, FrameArgs fr) {
    const uint32_t tile_cap = TILE ? TILE : tile_cap_, L = LINES ? LINES : L_;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ unsigned long long pacc[PROF ? 10 : 1];
    GelfFormat fmt{smem + tile_cap + 64u + (tile_cap / 16u + 16u) * 2u * GelfFormat::kClasses, tile_cap, L, pacc};
    if (PROF && threadIdx.x < 10) pacc[threadIdx.x] = 0ull;
    {
        const gelf2::Lds lds = gelf2::carve(smem, reinterpret_cast<uint16_t*>(smem + tile_cap + 64u), tile_cap, fmt.extra, L);
        gelf2::init_lds(lds);
        gelf2::clear_dirty(lds, tile_cap);
    }
    // The tables' forty words live in LDS, not in scalar registers: with them the kernel needs more than the 102 it has, and the
    // compiler parks whole 16-register kernel-argument tuples in VGPR lanes -- every row store and every entry store then began
    // with sixteen or thirty-two v_readlane (7 % of the kernel's VALU instructions); a column pointer is now one LDS read where used.
    __shared__ DevTables t_lds;
    if (threadIdx.x == 0) t_lds = t;
    __syncthreads();
    persistent_loop<NB, PROF>(bytes, offsets, n, t_lds, tile_cap, L, groups, prof, stash_base, fmt, fr);
    if (PROF) {
        __syncthreads();
        if (threadIdx.x < 10) atomicAdd(&prof[6 + threadIdx.x], pacc[threadIdx.x]);
    }
}

// The fast form over a RAW stream: the kernel frames its tiles itself (fg_fused.hpp); the lines it hands back get kPending as ever,
// and k_gelf_general finds them through the offsets this kernel wrote.
template <int NB, int MINW = 4, uint32_t TILE = 0, uint32_t LINES = 0>
__global__ __launch_bounds__(kWave, MINW) void k_gelf_fused(const uint8_t* __restrict__ bytes, DevTables t, uint32_t tile_cap_, uint32_t L_,
                                                         FusedArgs fa, uint32_t strip) {
    const uint32_t tile_cap = TILE ? TILE : tile_cap_, L = LINES ? LINES : L_;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    GelfFormat fmt{smem + tile_cap + 64u + (tile_cap / 16u + 16u) * 2u * GelfFormat::kClasses, tile_cap, L, nullptr};
    {
        const gelf2::Lds lds = gelf2::carve(smem, reinterpret_cast<uint16_t*>(smem + tile_cap + 64u), tile_cap, fmt.extra, L);
        gelf2::init_lds(lds);
        gelf2::clear_dirty(lds, tile_cap);
    }
    __shared__ DevTables t_lds;  // (see k_gelf)
    __shared__ FusedArgs fa_lds;
    if (threadIdx.x == 0) {
        t_lds = t;
        fa_lds = fa;
    }
    __syncthreads();
    fused_loop<NB>(bytes, t_lds, tile_cap, L, fmt, fa_lds, strip, nullptr);
}

// ---- kernel 2: pending lines -> the general form, straight from global memory --------------------------------------
// lines a wave of the general kernel collects pending lines from at a time -- at most; a small batch takes shorter spans (the
// launcher's `span`: a multiple of 64).  A wave runs the SHAPES of its pending lines one after the other (each a chain of dependent
// loads: DESIGN 3.3), so the work-efficient span is a long one -- full trips -- and the latency-efficient one is short: few shapes
// share a wave.  A batch that cannot fill the grid anyway takes the short ones: launches of 4 K / 16 K / 64 K / 256 K lines 118 / 120 /
// 185 / 307 -> 85 / 89 / 155 / 277 us, 4 M lines and more unchanged (profiles/r05ak_small_gelf_span.log).
constexpr uint32_t kGeneralSpan = 4096;
constexpr uint32_t kLaneBlock = kMaxStored * 4u + kMaxDepth / 8u;  // per lane: keypos[32] + the nesting stack (kMaxDepth bits)

// (Round 5, measured and dropped: the pending lines of a trip staged into LDS by the wave and walked from there -- 4 K .. 64 K-line
//  launches 105 / 109 / 173 us against 115 / 119 / 185 us, 16 M lines 2.00 against 2.05 G lines/s because 40 KiB of slots leave two
//  waves per CU: profiles/r05aa_gelf_general_lds_ab.log.  The ~70 us a trip takes however few lines it holds are not memory: a lane
//  walks its line three times -- validation, the sorted dispatch's count, its emission -- at ~20 dependent instructions per byte.)
__global__ __launch_bounds__(kWave, 2) void k_gelf_general(const uint8_t* __restrict__ bytes, const uint64_t* __restrict__ offsets,
                                                          uint64_t n, DevTables t, FrameArgs fr, uint32_t span, const unsigned long long* n_dev) {
    // (behind a fused launch -- fg_fused.hpp -- the number of frames is known on the device only: n is then the tables' capacity)
    if (n_dev) {
        const unsigned long long nd = *n_dev;
        if (nd < n) n = nd;
    }
    __shared__ __attribute__((aligned(16))) uint8_t scratch[kWave * kLaneBlock];
    __shared__ uint32_t ent_state[2];
    const uint32_t lane = threadIdx.x;
    if (lane < 2u) ent_state[lane] = 0u;
    __syncthreads();
    uint32_t* keypos = reinterpret_cast<uint32_t*>(scratch + lane * kLaneBlock);
    uint32_t* stack = reinterpret_cast<uint32_t*>(scratch + lane * kLaneBlock + kMaxStored * 4u);
    // (the fast form says whether it left anything at all: one word of the ctx's ring per launch)
    if (t.pending && *t.pending != t.epoch) return;
    // Pending lines are rare (~1 per 100): a wave first COLLECTS the pending lines of a 4096-line span into an LDS list, then
    // takes them 64 at a time -- lane per pending line instead of one busy lane per 64-line chunk.
    __shared__ uint16_t s_list[kGeneralSpan];
    const uint64_t spans = (n + span - 1) / span;  // (span: 64 .. kGeneralSpan, a multiple of 64)
    for (uint64_t sp = blockIdx.x; sp < spans; sp += gridDim.x) {
      const uint64_t l0 = sp * span;
      uint32_t cnt = 0;  // wave-uniform
      for (uint32_t k0 = 0; k0 < span / kWave; k0 += 8u) {
          uint32_t m8[8];
#pragma unroll
          for (uint32_t j = 0; j < 8u; ++j) {  // (eight loads in flight; rows beyond the span read as "not pending")
              const uint64_t q = l0 + (uint64_t)(k0 + j) * kWave + lane;
              m8[j] = (k0 + j) * kWave < span && q < n ? t.meta[q] : 0u;
          }
#pragma unroll
          for (uint32_t j = 0; j < 8u; ++j) {
              const bool p = (m8[j] & 0xFFu) == kPending;
              const unsigned long long b = __ballot(p);
              if (p) s_list[cnt + (uint32_t)__popcll(b & ((1ull << lane) - 1ull))] = (uint16_t)((k0 + j) * kWave + lane);
              cnt += (uint32_t)__popcll(b);
          }
      }
      __syncthreads();
      for (uint32_t at = 0; at < cnt; at += kWave) {
        const bool mine = at + lane < cnt;
        const uint64_t li = l0 + (mine ? (uint32_t)s_list[at + lane] : 0u);
        GRow r;
        uint64_t o0 = 0;
        uint32_t len = 0;
        if (mine) {
            o0 = offsets[li];
            uint64_t e1 = offsets[li + 1];
            // terminator stripping (BufRead::lines / split(0) semantics, as in the pipeline)
            if (fr.strip != FG_FRAME_NONE && e1 > o0) {
                const uint32_t b1 = bytes[e1 - 1];
                if (fr.strip == FG_FRAME_LINE) {
                    if (b1 == '\n') {
                        --e1;
                        if (e1 > o0 && bytes[e1 - 1] == '\r') --e1;
                    }
                } else if (b1 == 0u) {
                    --e1;
                }
            }
            len = (uint32_t)(e1 - o0);
            GlobalReader rd(reinterpret_cast<const uint32_t*>(bytes), o0);
            gelf_line(rd, len, keypos, stack, r, t);
            if (r.status != G_OK) r.n_ent = 0;
        }
        bool overflow;
        const uint32_t first = alloc_entries(t, mine ? r.n_ent : 0u, &overflow, ent_state);
        if (mine && overflow) {
            r.status = FG_ST_OVERFLOW;
            r.n_ent = 0;
        }
        if (mine && r.n_ent != 0) {
            GRow tmp = r;
            GlobalReader rd(reinterpret_cast<const uint32_t*>(bytes), o0);
            Gelf<GlobalReader> g{rd, len, (r.flags & FG_F_GELF_RETRY) != 0, stack};
            gelf_sorted_dispatch<true>(g, keypos, tmp, t, first);
        }
        if (mine) {
            RowOut o;
            const fg_span none{0, FG_NONE};
            const bool ok = r.status == G_OK;
            o.meta = r.status | (0xFFu << 8) | ((ok ? r.severity : 0xFFu) << 16) | ((ok ? r.flags : 0u) << 24);
            o.ts = (ok && r.have_ts) ? r.ts : 0.0;
            o.span[S_HOST] = ok ? fg_span{r.host_off, r.host_len} : none;
            o.span[S_APP] = none;
            o.span[S_PROC] = none;
            o.span[S_MSGID] = none;
            o.span[S_MSG] = ok ? fg_span{r.msg_off, r.msg_len} : none;
            o.span[S_FULL] = ok ? fg_span{r.full_off, r.full_len} : none;
            o.first = r.n_ent ? first : 0u;
            o.count = r.n_ent;
            store_row(t, li, o);
        }
      }
      __syncthreads();  // the list is rebuilt for the next span
    }
}

}  // namespace fg

namespace {
// the fast-form kernel with a register window of NB KiB (the bytes of the NEXT group, prefetched while this one is decoded)
template <int NB>
int launch_gelf_fast(const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t n, const fg::DevTables* t, uint64_t avg_len,
                     hipStream_t stream, uint32_t max_lines, fg::FrameArgs fr, const fg_launch_opts& lo, fg::TicketSlot* tk) {
    fg::LaunchPlan p;
    if (fg::plan_launch(fg::k_gelf<NB, false>, n, avg_len, 0u, 40960u, 0u, &p, lo, fg::PlanFormat().lines(max_lines).classes(fg::GelfFormat::kClasses).lds_for(fg::gelf_extra_lds)))
        return -1;
    dim3 grid(p.blocks), block(fg::kWave);
    fg::DevTables tt = *t;
#if defined(FG_PROF_BUILD)
    if (getenv("FG_PLAN")) fprintf(stderr, "gelf plan: L %u tile %u lds %u blocks %u window %d KiB\n", p.L, p.tile, p.lds, p.blocks, NB);
#endif
    if constexpr (NB == 3) {
        if (p.tile == 4096u && p.L == 8u && !fg::prof_requested() && !(lo.flags & FG_LO_GELF_GENERIC)) {
            // the geometry of ~300-byte GELF (the BASELINE corpus): constants (every LDS address an immediate)
            if (!lo.tile_cap && 8u * avg_len * 17u / 16u + 256u <= 3072u) {
                // ... and when the average group of eight fits 3 KiB, a tile of exactly the register window: 6.3 KiB of LDS and 96
                // registers (two spilled) let twenty waves share a CU (4 KiB tiles: 8 040 B, and only eighteen fit -- 1481 vs
                // 1422 M lines/s at sixteen, profiles/r04s_sweep_cfg3.log)
                fg_launch_opts lo3 = lo;
                lo3.tile_cap = 3072u;
                // (chunks of 64 lines = eight tiles, drawn by ticket: 4 M lines 1878-1886 M lines/s against 1805-1813 at 128, 1866-1870 at 48 / 96;
                //  16 M lines 2051 against 2012 at 128 and 2003 at 256 -- one box, alternated, profiles/r05u_chunk_taper_sweep.log;
                //  tickets from four chunks per wave on: at three -- 1 M lines -- one share per wave is as fast, profiles/r05v_policy_ab.log)
                if (fg::plan_launch(fg::k_gelf<NB, false, 5, 3072u, 8u>, n, avg_len, 0u, 40960u, 0u, &p, lo3,
                                    fg::PlanFormat().lines(max_lines).classes(fg::GelfFormat::kClasses).lds_for(fg::gelf_extra_lds).chunk(64u).tickets_from(4u)) ||
                    p.tile != 3072u || p.L != 8u)
                    return -1;
                fg::take_tickets(&fr, tk, p);
    tt.alloc_chunk = fg::entry_chunk(tt.ent_cap, p.blocks, n, lo, tt.shares);
                hipLaunchKernelGGL((fg::k_gelf<NB, false, 5, 3072u, 8u>), dim3(p.blocks), block, p.lds, stream, d_bytes, d_offsets, n, tt,
                                   p.tile, p.L, p.chunk, (unsigned long long*)nullptr, (uint64_t*)nullptr, fr);
                return 0;
            }
            fg::take_tickets(&fr, tk, p);
    tt.alloc_chunk = fg::entry_chunk(tt.ent_cap, p.blocks, n, lo, tt.shares);
            hipLaunchKernelGGL((fg::k_gelf<NB, false, 4, 4096u, 8u>), grid, block, p.lds, stream, d_bytes, d_offsets, n, tt, p.tile, p.L,
                               p.chunk, (unsigned long long*)nullptr, (uint64_t*)nullptr, fr);
            return 0;
        }
    }
#if defined(FG_PROF_BUILD)
    if (fg::prof_requested()) {
        fg::ProfRun pr;
        if (!pr.begin(stream)) return -1;
        tt.alloc_chunk = fg::entry_chunk(tt.ent_cap, p.blocks, n, lo, tt.shares);
        hipLaunchKernelGGL((fg::k_gelf<NB, true>), grid, block, p.lds, stream, d_bytes, d_offsets, n, tt, p.tile, p.L, p.chunk, pr.d,
                           (uint64_t*)nullptr, fr);
        pr.end(stream, "gelf", p);
        return 0;
    }
#endif
    fg::take_tickets(&fr, tk, p);
    tt.alloc_chunk = fg::entry_chunk(tt.ent_cap, p.blocks, n, lo, tt.shares);
    hipLaunchKernelGGL((fg::k_gelf<NB, false>), grid, block, p.lds, stream, d_bytes, d_offsets, n, tt, p.tile, p.L, p.chunk,
                       (unsigned long long*)nullptr, (uint64_t*)nullptr, fr);
    return 0;
}
}  // namespace

extern "C" int fg_launch_gelf_general(const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t n, const fg::DevTables* t, hipStream_t stream,
                                      uint32_t strip, const uint8_t* line_bad);
extern "C" int fg_launch_gelf(const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t n, const fg::DevTables* t,
                              uint64_t avg_len, hipStream_t stream, uint64_t* stash, uint32_t stash_blocks, uint32_t strip,
                              const uint8_t* line_bad, const fg_launch_opts* lop, fg::TicketSlot* tk) {
    const fg_launch_opts& lo = *lop;
    (void)stash;
    (void)stash_blocks;
    if (n == 0) return 0;
    fg::LaunchPlan p;
    // The fast form keeps its records in registers and writes entries straight to the table: what it needs in LDS is the tile,
    // four class bitmaps and the per-item / per-line arrays, which grow with the lines per group.  Take the largest power of two
    // whose LDS still lets sixteen waves share a CU (measured on the 307-byte corpus, M lines/s: 8 lines per group 1050,
    // 16: 677, 32: 803, 64: 398, 4: 436).  No global stash, no cap on the grid.
    uint32_t max_lines = fg::kGelfLines;
    uint32_t lds_budget = 10u * 1024u;  // sixteen waves per CU
    if (lo.gelf_lds_budget) lds_budget = lo.gelf_lds_budget;  // tuning
    if (lo.lines_per_group) max_lines = lo.lines_per_group;
    else {
        while (max_lines > 4u) {
            if (fg::plan_launch(fg::k_gelf<2, false>, n, avg_len, 0u, 40960u, 0u, &p, lo, fg::PlanFormat().lines(max_lines).classes(fg::GelfFormat::kClasses).lds_for(fg::gelf_extra_lds)))
                return -1;
            if (p.L < max_lines) max_lines = p.L;  // (the geometry already settled on fewer lines)
            if (p.lds <= lds_budget) break;
            max_lines >>= 1;
        }
    }
    if (fg::plan_launch(fg::k_gelf<2, false>, n, avg_len, 0u, 40960u, 0u, &p, lo, fg::PlanFormat().lines(max_lines).classes(fg::GelfFormat::kClasses).lds_for(fg::gelf_extra_lds)))
        return -1;
    // The register window should hold the whole average group: what lies beyond it is staged by plain loads whose latency
    // nothing hides (and four chunks at a time).  1 KiB of window = 4 registers.
    uint64_t want = ((uint64_t)p.L * avg_len * 17u / 16u + 128u + 1023u) / 1024u;
    if (lo.gelf_window_kib) want = lo.gelf_window_kib;  // tuning
    const fg::FrameArgs fr{strip, line_bad};
    const dim3 block(fg::kWave);
    int rc;
    if (want <= 2) rc = launch_gelf_fast<2>(d_bytes, d_offsets, n, t, avg_len, stream, p.L, fr, lo, tk);
    else if (want == 3) rc = launch_gelf_fast<3>(d_bytes, d_offsets, n, t, avg_len, stream, p.L, fr, lo, tk);
    else if (want == 4) rc = launch_gelf_fast<4>(d_bytes, d_offsets, n, t, avg_len, stream, p.L, fr, lo, tk);
    else if (want == 5) rc = launch_gelf_fast<5>(d_bytes, d_offsets, n, t, avg_len, stream, p.L, fr, lo, tk);
    else rc = launch_gelf_fast<6>(d_bytes, d_offsets, n, t, avg_len, stream, p.L, fr, lo, tk);
    if (rc) return rc;
    if (hipGetLastError() != hipSuccess) return -1;
    // pending lines (a frame flagged as invalid UTF-8 never is: the pipeline has overwritten its status) -- unless the caller runs the
    // exact form ONCE for all the slices of a batch (fg_launch_gelf_general over all rows: the kernel is a chain of dependent steps,
    // ~170 us however few lines a slice hands it -- a quarter of the GPU time of a sliced host path, profiles/r04g2_probe_frame_gelf.log)
    if (lo.flags & FG_LO_RESERVED) return 0;
    return fg_launch_gelf_general(d_bytes, d_offsets, n, t, stream, strip, line_bad);
}
extern "C" int fg_launch_gelf_general_dev(const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t n, const fg::DevTables* t, hipStream_t stream,
                                          uint32_t strip, const uint8_t* line_bad, const unsigned long long* n_dev);
extern "C" int fg_launch_gelf_general(const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t n, const fg::DevTables* t, hipStream_t stream,
                                      uint32_t strip, const uint8_t* line_bad) {
    return fg_launch_gelf_general_dev(d_bytes, d_offsets, n, t, stream, strip, line_bad, nullptr);
}
// n_dev != null: the rows are min(n, *n_dev), read on the device (the exact form behind a fused launch)
extern "C" int fg_launch_gelf_general_dev(const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t n, const fg::DevTables* t, hipStream_t stream,
                                          uint32_t strip, const uint8_t* line_bad, const unsigned long long* n_dev) {
    if (n == 0) return 0;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1)
        return -1;
    // the span: as long as it can be with a span for every wave of the grid (n / 1024 lines, a multiple of 64 in 64 .. 4096)
    uint64_t gblocks = (uint64_t)cus * 8u;
    uint32_t span = fg::kGeneralSpan;
    while (span > fg::kWave && n < (uint64_t)span * 1024u) span >>= 1;
    const uint64_t chunks = (n + span - 1) / span;
    if (gblocks > chunks) gblocks = chunks;
    hipLaunchKernelGGL(fg::k_gelf_general, dim3((uint32_t)gblocks), dim3(fg::kWave), 0, stream, d_bytes, d_offsets, n, *t, fg::FrameArgs{strip, line_bad},
                       span, n_dev);
    return (int)hipGetLastError();
}

// The fused launch (fg_fused.hpp): frame + the fast form of a raw stream chunk in one kernel (see fg_launch_rfc5424_fused).  The lines
// the fast form hands back are finished by fg_launch_gelf_general_dev over the offsets this launch writes (n read on the device).
// waves per SIMD the resident fused instantiation is compiled for (A/B builds: -DFG_GELF_FUSED_MINW=2 / 4).  Measured on one box
// (profiles/r06ar_gelf_fused_minw.log, 4 M lines): two waves (228 VGPRs, no scratch) 5.39 ms, three (168, 144 bytes of scratch) 5.12,
// four (128, 272 bytes) 5.35 -- the spills are not what holds this kernel back; across the link four waves (384 bytes) gave 0.82 of the
// link and two (no scratch) 0.79, so that instantiation stays at four.
#ifndef FG_GELF_FUSED_MINW
#define FG_GELF_FUSED_MINW 3
#endif
extern "C" int fg_launch_gelf_fused(const uint8_t* d_bytes, uint64_t nbytes, const fg::DevTables* t, const fg::FusedGeom* g, hipStream_t stream,
                                    uint32_t strip, int final_, uint64_t* d_offsets, uint64_t cap, uint8_t* scratch, const fg_launch_opts* lo,
                                    unsigned long long** d_total) {
    if (nbytes == 0 || !g->ok) return -1;
    const uint32_t base_lds = g->tile + 64u + (g->tile / 16u + 16u) * 2u * fg::GelfFormat::kClasses + fg::gelf_extra_lds(g->tile, g->L);
    fg::FusedArgs fa{};
    uint32_t lds = 0, blocks = 0;
    const uint32_t delim = strip == FG_FRAME_LINE ? 0x0Au : 0u;
    const bool konst = g->variant == 1u && g->tile == 3072u && g->L == 8u;
    const int prc = konst ? fg::fused_prepare(fg::k_gelf_fused<3, FG_GELF_FUSED_MINW, 3072u, 8u>, *g, base_lds, nbytes, final_, delim, d_offsets, cap, scratch, 0u, *lo, stream,
                                              &fa, &lds, &blocks)
                          : fg::fused_prepare(fg::k_gelf_fused<6, 4>, *g, base_lds, nbytes, final_, delim, d_offsets, cap, scratch, 0u, *lo, stream, &fa, &lds,
                                              &blocks);
    if (prc) return -1;
    fg::DevTables tt = *t;
    tt.alloc_chunk = fg::entry_chunk(tt.ent_cap, blocks, nbytes / (g->S / (g->plan ? g->plan : g->L) ? g->S / (g->plan ? g->plan : g->L) : 1u) + 1u, *lo, tt.shares);
    *d_total = fa.total;
    if (konst)
        hipLaunchKernelGGL((fg::k_gelf_fused<3, FG_GELF_FUSED_MINW, 3072u, 8u>), dim3(blocks), dim3(fg::kWave), lds, stream, d_bytes, tt, g->tile, g->L, fa, strip);
    else
        hipLaunchKernelGGL((fg::k_gelf_fused<6, 4>), dim3(blocks), dim3(fg::kWave), lds, stream, d_bytes, tt, g->tile, g->L, fa, strip);
    return (int)hipGetLastError();
}
