// Host build of flowgger_amd/csrc/fg_emit.hpp + fg_enc_cfg.hpp (the encoder emitters and their configuration exactly
// as the kernels / the C ABI use them), driven from a canonical Record: the record's strings are laid out in a
// synthetic "line" (optionally escaped the way a decoder's source text would be: RFC5424 SD escapes, JSON escapes)
// and a one-row table with spans + flags points into it.  Test infrastructure only -- the product has no CPU path.
#include <cstdint>
#include <cstring>
#include <optional>
#include <string>
#include <vector>

#include "../../flowgger_amd/csrc/fg_enc_cfg.hpp"

namespace {
struct HostReader {
    const uint8_t* p;
    uint32_t byte(uint32_t i) { return p[i]; }
    uint32_t load4(uint32_t i, uint32_t nb) {
        uint32_t w = 0;
        memcpy(&w, p + i, nb);  // never reads past the wanted bytes: the line buffer has no padding
        return w;
    }
    void load16(uint32_t i, uint32_t* q) { memcpy(q, p + i, 16); }  // all 16 bytes are wanted by contract
    void load16p(uint32_t i, uint32_t nb, uint32_t* q) {  // the first nb wanted, the rest unspecified: garbage, so that a missing mask shows
        memset(q, 0xEE, 16);
        memcpy(q, p + i, nb);
    }
};
struct Cur {
    const uint8_t* p;
    uint64_t n, i = 0;
    bool ok = true;
    uint32_t u8() { if (i + 1 > n) { ok = false; return 0; } return p[i++]; }
    uint32_t u32() { uint32_t v = 0; if (i + 4 > n) { ok = false; return 0; } memcpy(&v, p + i, 4); i += 4; return v; }
    uint64_t u64() { uint64_t v = 0; if (i + 8 > n) { ok = false; return 0; } memcpy(&v, p + i, 8); i += 8; return v; }
    std::string str() { uint32_t l = u32(); if (i + l > n) { ok = false; return ""; } std::string s((const char*)p + i, l); i += l; return s; }
};
uint64_t rng_state;
uint32_t g_sort_slots = fg::emit::kSortSlots;
uint32_t g_last_plain = 0;
uint32_t rnd() {
    rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull;
    return (uint32_t)(rng_state >> 33);
}
// text -> the escaped source form; *esc = whether the span needs the unescape flag
std::string sd_escape(const std::string& s, bool* esc) {
    std::string o;
    *esc = false;
    for (char c : s) {
        if (c == '"' || c == '\\' || c == ']') { o.push_back('\\'); *esc = true; }
        o.push_back(c);
    }
    return o;
}
std::string json_escape_src(const std::string& s, bool* esc) {
    static const char hex[] = "0123456789abcdefABCDEF";
    std::string o;
    *esc = false;
    for (size_t i = 0; i < s.size();) {
        uint8_t c = (uint8_t)s[i];
        auto u4 = [&](uint32_t v) {
            o += "\\u";
            for (int k = 3; k >= 0; --k) {
                uint32_t h = (v >> (4 * k)) & 15;
                o.push_back(h >= 10 && (rnd() & 1) ? hex[h + 6] : hex[h]);
            }
        };
        if (c == '"' || c == '\\') { o.push_back('\\'); o.push_back((char)c); *esc = true; ++i; continue; }
        if (c < 0x20) {
            *esc = true;
            const char* sh = c == 8 ? "\\b" : c == 9 ? "\\t" : c == 10 ? "\\n" : c == 12 ? "\\f" : c == 13 ? "\\r" : nullptr;
            if (sh && (rnd() & 1)) o += sh;
            else u4(c);
            ++i;
            continue;
        }
        if (c == '/' && (rnd() & 1)) { o += "\\/"; *esc = true; ++i; continue; }
        if (c < 0x80) {
            if ((rnd() & 15) == 0) { u4(c); *esc = true; }
            else o.push_back((char)c);
            ++i;
            continue;
        }
        // a UTF-8 sequence: raw, or as \uXXXX (a surrogate pair above the BMP)
        int len = c >= 0xF0 ? 4 : c >= 0xE0 ? 3 : 2;
        if (i + len > s.size()) len = (int)(s.size() - i);
        if (rnd() & 1) {
            uint32_t cp = len == 4 ? (c & 7) : len == 3 ? (c & 15) : (c & 31);
            for (int k = 1; k < len; ++k) cp = (cp << 6) | ((uint8_t)s[i + k] & 0x3F);
            if (cp >= 0x10000) {
                cp -= 0x10000;
                u4(0xD800 + (cp >> 10));
                u4(0xDC00 + (cp & 0x3FF));
            } else {
                u4(cp);
            }
            *esc = true;
        } else {
            o.append(s, i, len);
        }
        i += len;
    }
    return o;
}
}  // namespace

// the GELF ranking scratch size the kernels pick per batch (8 or 32)
extern "C" void fge_set_sort_slots(uint32_t n) { g_sort_slots = n; }
// the count pass's note for the write pass on the last record encoded: 1 = no span held a byte to escape
extern "C" uint32_t fge_last_plain() { return g_last_plain; }

// enc / merger: fg_encoder / fg_merger.  src_fmt: which decoder the synthetic row pretends to come from (escape style).
// gelf_key_variant (src_fmt == FG_GELF): 0 = key spans include the leading '_', 1 = they do not (the decoder adds it).
// Returns the output length (written when <= cap), -1 = bad canonical record, -2 = key without '_' ; *status = encode status.
extern "C" int64_t fge_encode_canonical(int enc, int merger, int src_fmt, int gelf_key_variant, uint64_t seed, const uint8_t* canonical,
                                        uint64_t len, const char* const* extra_keys, const char* const* extra_vals, uint32_t n_extra,
                                        const char* prepend, double now_ts, uint8_t* out, uint64_t cap, uint32_t* status) {
    rng_state = seed * 2654435761u + 12345u;
    Cur c{canonical, len};
    if (c.u8() != 0) return -1;
    const bool ts_now = c.u8() != 0;
    uint64_t tsb = c.u64();
    const uint32_t fac = c.u8(), sev = c.u8();
    std::string line(16, 'x');  // spans never start at 0: catches base mistakes
    uint32_t flags = ts_now ? FG_F_TS_NOW : 0;
    fg_span cols[6];
    const uint32_t escbit[6] = {FG_F_HOST_ESC, 0, 0, 0, FG_F_MSG_ESC, FG_F_FULLMSG_ESC};
    for (int k = 0; k < 6; ++k) {
        if (!c.u8()) { cols[k] = fg_span{0, FG_NONE}; continue; }
        std::string s = c.str();
        bool esc = false;
        if (src_fmt == FG_GELF && escbit[k]) s = json_escape_src(s, &esc);
        if (esc) flags |= escbit[k];
        line += "|";
        cols[k] = fg_span{(uint32_t)line.size(), (uint32_t)s.size()};
        line += s;
    }
    std::vector<fg_span> en;
    std::vector<uint64_t> ev;
    std::vector<uint8_t> et, ef;
    if (c.u8()) {
        const uint32_t nsd = c.u32();
        for (uint32_t a = 0; a < nsd; ++a) {
            const bool has_id = c.u8() != 0;
            std::string id = has_id ? c.str() : "";
            if (src_fmt == FG_RFC5424) {  // every element has an sd_id header entry
                line += "[";
                en.push_back(fg_span{(uint32_t)line.size(), (uint32_t)id.size()});
                line += id;
                ev.push_back(0);
                et.push_back(FG_T_SDID);
                ef.push_back(0);
            }
            const uint32_t np = c.u32();
            for (uint32_t b = 0; b < np; ++b) {
                std::string key = c.str();
                const uint32_t ty = c.u8();
                uint8_t fl = 0;
                if (key.empty() || key[0] != '_') return -2;
                std::string name = (src_fmt == FG_GELF && gelf_key_variant == 0) ? key : key.substr(1);
                if (src_fmt == FG_GELF && gelf_key_variant == 1 && !name.empty() && name[0] == '_') name = key;  // "__x" can only come from "__x"
                bool esc = false;
                if (src_fmt == FG_GELF) name = json_escape_src(name, &esc);
                if (esc) fl |= FG_EF_NAME_ESC;
                line += " ";
                en.push_back(fg_span{(uint32_t)line.size(), (uint32_t)name.size()});
                line += name;
                uint64_t v = 0;
                if (ty == FG_T_STRING) {
                    std::string s = c.str();
                    bool vesc = false;
                    if (src_fmt == FG_RFC5424) s = sd_escape(s, &vesc);
                    else if (src_fmt == FG_GELF) s = json_escape_src(s, &vesc);
                    if (vesc) fl |= FG_EF_VAL_ESC;
                    line += "=";
                    v = (uint64_t)line.size() | ((uint64_t)s.size() << 32);
                    line += s;
                } else if (ty == FG_T_BOOL) {
                    v = c.u8();
                } else if (ty != FG_T_NULL) {
                    v = c.u64();
                }
                ev.push_back(v);
                et.push_back((uint8_t)ty);
                ef.push_back(fl);
            }
        }
    }
    if (!c.ok) return -1;
    line += "  tail";
    // the one-row table (row index 3 of 5, entries at an offset: catches indexing mistakes)
    const uint64_t li = 3;
    std::vector<uint32_t> meta(5, 0xFFFFFFFFu), ent_first(5, 0), ent_count(5, 0);
    std::vector<double> ts(5, 0.0);
    std::vector<fg_span> span[6];
    for (int k = 0; k < 6; ++k) {
        span[k].assign(5, fg_span{0, FG_NONE});
        span[k][li] = cols[k];
    }
    meta[li] = 0u | fac << 8 | sev << 16 | flags << 24;
    memcpy(&ts[li], &tsb, 8);
    const uint32_t base = 7;
    std::vector<fg_span> ent_name(base + en.size() + 1, fg_span{0, 0});
    std::vector<uint64_t> ent_val(base + en.size() + 1, 0);
    std::vector<uint8_t> ent_type(base + en.size() + 1, 0), ent_flags(base + en.size() + 1, 0);
    for (size_t k = 0; k < en.size(); ++k) {
        ent_name[base + k] = en[k];
        ent_val[base + k] = ev[k];
        ent_type[base + k] = et[k];
        ent_flags[base + k] = ef[k];
    }
    ent_first[li] = base;
    ent_count[li] = (uint32_t)en.size();
    fg::DevTables t{};
    t.n = 5;
    t.ent_cap = ent_name.size();
    t.meta = meta.data();
    t.ts = ts.data();
    for (int k = 0; k < 6; ++k) t.span[k] = span[k].data();
    t.ent_first = ent_first.data();
    t.ent_count = ent_count.data();
    t.ent_name = ent_name.data();
    t.ent_val = ent_val.data();
    t.ent_type = ent_type.data();
    t.ent_flags = ent_flags.data();

    fg_encode_cfg ec{};
    ec.encoder = (fg_encoder)enc;
    ec.merger = (fg_merger)merger;
    ec.n_extra = n_extra;
    ec.extra_keys = extra_keys;
    ec.extra_values = extra_vals;
    ec.prepend = prepend;
    ec.now_ts = now_ts;
    const std::string suffix[4];
    const bool has_suffix[4] = {false, false, false, false};
    fg::EncCfgHost h;
    if (!fg::build_enc_cfg((fg_format)src_fmt, &ec, suffix, has_suffix, &h)) return -1;
    h.cfg.blob = h.blob.data();
    h.cfg.sort_slots = g_sort_slots;
    h.cfg.keys = h.keys.data();

    uint64_t keys64[fg::emit::kSortSlots];
    uint8_t slot_ent[fg::emit::kSortSlots], order[fg::emit::kSortSlots];
    HostReader rd{(const uint8_t*)line.data()};
    uint32_t st = 0, size = 0;
    std::vector<uint8_t> res;
    // the write pass runs through the kernels' PackSink (dword and 16-byte stores) at all sixteen start alignments, inside a guarded buffer: nothing
    // outside [start, start + size) may change
    for (uint32_t al = 0; al < 16; ++al) {
        std::vector<uint8_t> buf;
#define RUN(E)                                                                                                  \
    case E: {                                                                                                   \
        uint32_t plain = 0;                                                                                     \
        size = fg::emit::row_size<E>(h.cfg, rd, t, li, meta[li], keys64, slot_ent, order, &st, nullptr, &plain); \
        g_last_plain = plain;                                                                                   \
        buf.assign((size_t)size + 96, 0xA5);                                                                    \
        uint8_t* start = buf.data() + 32;                                                                       \
        start += (al - ((uintptr_t)start & 15u)) & 15u;                                                          \
        fg::emit::PackSink sink(start);                                                                         \
        /* odd alignments: with the row in registers and the count pass's "no byte to escape" note, as the write kernel runs */ \
        fg::emit::RowRegs pre;                                                                                  \
        pre.load(t, li);                                                                                        \
        pre.plain = plain;                                                                                      \
        fg::emit::row_write<E>(sink, size, h.cfg, rd, t, li, meta[li], keys64, slot_ent, order, (al & 1u) ? &pre : nullptr); \
        for (uint8_t* q = buf.data(); q < buf.data() + buf.size(); ++q)                                         \
            if ((q < start || q >= start + size) && *q != 0xA5) return -4;                                      \
        if (size && sink.p != start + size) return -3; /* count and write passes disagree */                   \
        std::vector<uint8_t> got(start, start + size);                                                          \
        if (al && got != res) return -5;                                                                        \
        res = got;                                                                                              \
        break;                                                                                                  \
    }
        switch (enc) {
            RUN(FG_ENC_GELF)
            RUN(FG_ENC_LTSV)
            RUN(FG_ENC_RFC5424)
            RUN(FG_ENC_RFC3164)
            RUN(FG_ENC_PASSTHROUGH)
            default: return -1;
        }
#undef RUN
    }
    if (status) *status = st;
    if (out && res.size() <= cap) memcpy(out, res.data(), res.size());
    return (int64_t)res.size();
}

// PackSink alone: a random sequence of put / put_word / put16 / put_part calls at a random start alignment into a guarded buffer
// must leave exactly the concatenation of the bytes it was handed, and nothing outside.  Returns 0, or the failing step + 1.
extern "C" int fge_sink_fuzz(uint64_t seed, uint32_t steps) {
    rng_state = seed * 0x9E3779B97F4A7C15ull + 777u;
    std::vector<uint8_t> want;
    std::vector<uint8_t> buf(steps * 16u + 128u, 0xA5);
    uint8_t* start = buf.data() + 48;
    start += (rnd() & 15u) - ((uintptr_t)start & 15u) & 15u;
    fg::emit::PackSink sink(start);
    for (uint32_t i = 0; i < steps; ++i) {
        uint32_t q[4];
        for (int k = 0; k < 4; ++k) q[k] = rnd() * 2654435761u + rnd();
        const uint32_t kind = rnd() % 4u;
        if (kind == 0u) {
            sink.put(q[0] & 0xFFu);
            want.push_back((uint8_t)q[0]);
        } else if (kind == 1u) {
            const uint32_t nb = 1u + rnd() % 4u;
            const uint32_t w = nb == 4u ? q[0] : (q[0] & ((1u << (8u * nb)) - 1u));
            sink.put_word(w, nb);
            for (uint32_t b = 0; b < nb; ++b) want.push_back((uint8_t)(w >> (8u * b)));
        } else if (kind == 2u) {
            sink.put16(q[0], q[1], q[2], q[3]);
            for (uint32_t b = 0; b < 16u; ++b) want.push_back((uint8_t)(q[b >> 2] >> (8u * (b & 3u))));
        } else {
            const uint32_t nb = 1u + rnd() % 16u;
            for (uint32_t b = nb; b < 16u; ++b) q[b >> 2] &= ~(0xFFu << (8u * (b & 3u)));  // (the contract: zero beyond nb)
            sink.put_part(q[0], q[1], q[2], q[3], nb);
            for (uint32_t b = 0; b < nb; ++b) want.push_back((uint8_t)(q[b >> 2] >> (8u * (b & 3u))));
        }
    }
    sink.finish();
    if (sink.p != start + want.size()) return -1;
    if (memcmp(start, want.data(), want.size()) != 0) return -2;
    for (uint8_t* p = buf.data(); p < buf.data() + buf.size(); ++p)
        if ((p < start || p >= start + want.size()) && *p != 0xA5) return -3;
    return 0;
}
