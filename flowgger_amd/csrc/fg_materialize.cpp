// fg_materialize.cpp -- host-side Record materialisation from decode tables.
//
// A flowgger `Record` owns one heap string per field (src/flowgger/record.rs:70-82), so building
// it is inherently a host-side copy; the GPU's product is the table row that says WHERE every
// field is.  This file turns rows + line bytes into the canonical Record serialisation that
// INTEGRATION.md documents (and that the Rust shim there mirrors when it fills a real `Record`).
// The only transformations besides copying are the ones the decoders apply while copying:
//   * "_" key prefix               rfc5424_decoder.rs:220-227, ltsv_decoder.rs:128-135,
//                                  gelf_decoder.rs:99-103 (GELF: only when missing)
//   * unescape_sd_value            rfc5424_decoder.rs:105-125   (flagged values only)
//   * JSON string unescaping       serde_json 0.8 read.rs parse_escape (flagged spans only)
//   * LTSV type suffix             ltsv_decoder.rs:131-136      (flagged entries only)
#include <cstdint>
#include <cstring>
#include <string>

#include "../../include/fg_hip.h"

namespace {

struct Sink {
    uint8_t* out;
    uint64_t cap;
    uint64_t n = 0;
    void put(const void* p, size_t len) {
        if (out && n + len <= cap) memcpy(out + n, p, len);
        n += len;
    }
    void u8(uint8_t v) { put(&v, 1); }
    void u32(uint32_t v) { put(&v, 4); }
    void u64(uint64_t v) { put(&v, 8); }
    void patch32(uint64_t at, uint32_t v) {
        if (out && at + 4 <= cap) memcpy(out + at, &v, 4);
    }
};

int hexv(uint8_t c) {
    if (c >= '0' && c <= '9') return c - '0';
    if (c >= 'a' && c <= 'f') return c - 'a' + 10;
    if (c >= 'A' && c <= 'F') return c - 'A' + 10;
    return 0;
}
void push_utf8(std::string* s, uint32_t c) {
    if (c < 0x80) {
        s->push_back((char)c);
    } else if (c < 0x800) {
        s->push_back((char)(0xC0 | (c >> 6)));
        s->push_back((char)(0x80 | (c & 0x3F)));
    } else if (c < 0x10000) {
        s->push_back((char)(0xE0 | (c >> 12)));
        s->push_back((char)(0x80 | ((c >> 6) & 0x3F)));
        s->push_back((char)(0x80 | (c & 0x3F)));
    } else {
        s->push_back((char)(0xF0 | (c >> 18)));
        s->push_back((char)(0x80 | ((c >> 12) & 0x3F)));
        s->push_back((char)(0x80 | ((c >> 6) & 0x3F)));
        s->push_back((char)(0x80 | (c & 0x3F)));
    }
}
// JSON escapes of an already-validated string body (the kernel rejected malformed ones).
void json_unescape(const uint8_t* p, uint32_t len, std::string* out, bool retry = false) {
    out->clear();
    for (uint32_t i = 0; i < len;) {
        uint8_t c = p[i];
        if (c != '\\' || i + 1 >= len) {
            out->push_back((char)c);
            ++i;
            continue;
        }
        uint8_t e = p[i + 1];
        i += 2;
        if (retry && e == '\n') {  // the reference replaced LF by "\\n": escaped backslash, then 'n'
            out->push_back('\\');
            out->push_back('n');
            continue;
        }
        switch (e) {
            case 'b': out->push_back('\x08'); break;
            case 'f': out->push_back('\x0c'); break;
            case 'n': out->push_back('\n'); break;
            case 'r': out->push_back('\r'); break;
            case 't': out->push_back('\t'); break;
            case 'u': {
                if (i + 4 > len) break;
                uint32_t n1 = hexv(p[i]) << 12 | hexv(p[i + 1]) << 8 | hexv(p[i + 2]) << 4 | hexv(p[i + 3]);
                i += 4;
                if (n1 >= 0xD800 && n1 <= 0xDBFF && i + 6 <= len) {
                    uint32_t n2 = hexv(p[i + 2]) << 12 | hexv(p[i + 3]) << 8 | hexv(p[i + 4]) << 4 | hexv(p[i + 5]);
                    i += 6;
                    n1 = (((n1 - 0xD800) << 10) | (n2 - 0xDC00)) + 0x10000;
                }
                push_utf8(out, n1);
                break;
            }
            default: out->push_back((char)e);  // " \ /
        }
    }
}
// byte length of the Unicode White_Space character at p[i] (0: none)
uint32_t ws_len(const uint8_t* p, uint32_t i, uint32_t n) {
    const uint8_t c = p[i];
    if (c < 0x80) return (c == 32 || (c >= 9 && c <= 13)) ? 1 : 0;
    if (c == 0xC2 && i + 1 < n) return (p[i + 1] == 0x85 || p[i + 1] == 0xA0) ? 2 : 0;
    if (i + 2 >= n) return 0;
    const uint8_t b1 = p[i + 1], b2 = p[i + 2];
    if (c == 0xE2) {
        if (b1 == 0x80) return ((b2 >= 0x80 && b2 <= 0x8A) || b2 == 0xA8 || b2 == 0xA9 || b2 == 0xAF) ? 3 : 0;
        return (b1 == 0x81 && b2 == 0x9F) ? 3 : 0;
    }
    if (c == 0xE1) return (b1 == 0x9A && b2 == 0x80) ? 3 : 0;
    if (c == 0xE3) return (b1 == 0x80 && b2 == 0x80) ? 3 : 0;
    return 0;
}
// str::split_whitespace(..).join(" ")
void ws_join(const uint8_t* p, uint32_t len, std::string* out) {
    out->clear();
    bool in_tok = false, any = false;
    for (uint32_t i = 0; i < len;) {
        const uint32_t w = ws_len(p, i, len);
        if (w) {
            in_tok = false;
            i += w;
            continue;
        }
        if (!in_tok) {
            if (any) out->push_back(' ');
            in_tok = any = true;
        }
        out->push_back((char)p[i]);
        ++i;
    }
}
// rfc5424_decoder.rs:105-125
void sd_unescape(const uint8_t* p, uint32_t len, std::string* out) {
    out->clear();
    bool esc = false;
    for (uint32_t i = 0; i < len; ++i) {
        char c = (char)p[i];
        if (!esc) {
            if (c == '\\') esc = true;
            else out->push_back(c);
        } else {
            if (c != '"' && c != '\\' && c != ']') out->push_back('\\');
            out->push_back(c);
            esc = false;
        }
    }
}

}  // namespace

extern "C" int64_t fg_tables_serialize(fg_format fmt, const fg_cfg* cfg, const uint8_t* bytes, const uint64_t* offsets,
                                       const fg_tables* t, uint64_t i0, uint64_t i1, uint8_t* out, uint64_t cap,
                                       uint64_t* out_offsets) {
    if (!t || !offsets || i1 < i0 || i1 > t->n) return FG_ERR_ARG;
    if ((int)fmt < 0 || (int)fmt > (int)FG_RFC3164) return FG_ERR_ARG;
    Sink k{out, cap};
    std::string tmp;
    const char* suffix[6] = {nullptr, cfg ? cfg->suffix_bool : nullptr, cfg ? cfg->suffix_f64 : nullptr,
                             cfg ? cfg->suffix_i64 : nullptr, cfg ? cfg->suffix_u64 : nullptr, nullptr};
    const fg_span* cols[6] = {t->hostname, t->appname, t->procid, t->msgid, t->msg, t->full_msg};
    const uint8_t esc_flag[6] = {FG_F_HOST_ESC, 0, 0, 0, FG_F_MSG_ESC, FG_F_FULLMSG_ESC};
    for (uint64_t i = i0; i < i1; ++i) {
        if (out_offsets) out_offsets[i - i0] = k.n;
        const uint8_t* line = bytes + offsets[i];
        const uint32_t m = t->meta[i];
        const uint8_t status = FG_META_STATUS(m);
        if (status != 0) {
            const char* e = fg_error_string(fmt, status);
            if (!e) e = status == FG_ST_OVERFLOW ? "<entry table overflow>" : "<unknown status>";
            k.u8(1);
            k.u32((uint32_t)strlen(e));
            k.put(e, strlen(e));
            continue;
        }
        const uint8_t flags = FG_META_FLAGS(m);
        k.u8(0);
        k.u8((flags & FG_F_TS_NOW) ? 1 : 0);
        uint64_t tb = 0;
        if (!(flags & FG_F_TS_NOW)) memcpy(&tb, &t->ts[i], 8);
        k.u64(tb);
        k.u8(FG_META_FACILITY(m));
        k.u8(FG_META_SEVERITY(m));
        for (int c = 0; c < 6; ++c) {
            fg_span s = cols[c][i];
            if (s.len == FG_NONE) {
                k.u8(0);
                continue;
            }
            k.u8(1);
            if (fmt == FG_GELF && (flags & esc_flag[c])) {
                json_unescape(line + s.off, s.len, &tmp, (flags & FG_F_GELF_RETRY) != 0);
                k.u32((uint32_t)tmp.size());
                k.put(tmp.data(), tmp.size());
            } else if (fmt == FG_RFC3164 && c == 4 && (flags & FG_F_MSG_JOIN)) {
                ws_join(line + s.off, s.len, &tmp);  // `_log_tokens[1..].join(" ")`, rfc3164_decoder.rs:70
                k.u32((uint32_t)tmp.size());
                k.put(tmp.data(), tmp.size());
            } else {
                k.u32(s.len);
                k.put(line + s.off, s.len);
            }
        }
        const uint32_t cnt = t->ent_count[i], first = t->ent_first[i];
        if (cnt == 0) {
            k.u8(0);
            continue;
        }
        k.u8(1);
        uint64_t n_sd_at = k.n;
        k.u32(0);
        uint32_t n_sd = 0, n_pairs = 0;
        uint64_t n_pairs_at = 0;
        if (fmt != FG_RFC5424) {  // one element, sd_id None (ltsv_decoder.rs:88,215; gelf_decoder.rs:35,119)
            n_sd = 1;
            k.u8(0);
            n_pairs_at = k.n;
            k.u32(0);
        }
        for (uint32_t e = first; e < first + cnt; ++e) {
            const fg_span nm = t->ent_name[e];
            const uint8_t ty = t->ent_type[e], ef = t->ent_flags[e];
            if (ty == FG_T_SDID) {
                if (n_sd) k.patch32(n_pairs_at, n_pairs);
                ++n_sd;
                n_pairs = 0;
                k.u8(1);
                k.u32(nm.len);
                k.put(line + nm.off, nm.len);
                n_pairs_at = k.n;
                k.u32(0);
                continue;
            }
            ++n_pairs;
            // key
            const uint8_t* np = line + nm.off;
            uint32_t nl = nm.len;
            std::string key;
            if (ef & FG_EF_NAME_ESC) {
                json_unescape(np, nl, &tmp, (flags & FG_F_GELF_RETRY) != 0);
                key = tmp;
            } else {
                key.assign((const char*)np, nl);
            }
            if (fmt != FG_GELF || key.empty() || key[0] != '_') key.insert(key.begin(), '_');
            if ((ef & FG_EF_SUFFIX) && ty <= FG_T_U64 && suffix[ty]) key += suffix[ty];
            k.u32((uint32_t)key.size());
            k.put(key.data(), key.size());
            k.u8(ty);
            const uint64_t v = t->ent_val[e];
            switch (ty) {
                case FG_T_STRING: {
                    const uint32_t vo = (uint32_t)v, vl = (uint32_t)(v >> 32);
                    if (ef & FG_EF_VAL_ESC) {
                        if (fmt == FG_RFC5424) sd_unescape(line + vo, vl, &tmp);
                        else json_unescape(line + vo, vl, &tmp, (flags & FG_F_GELF_RETRY) != 0);
                        k.u32((uint32_t)tmp.size());
                        k.put(tmp.data(), tmp.size());
                    } else {
                        k.u32(vl);
                        k.put(line + vo, vl);
                    }
                    break;
                }
                case FG_T_BOOL: k.u8((uint8_t)v); break;
                case FG_T_NULL: break;
                default: k.u64(v);
            }
        }
        k.patch32(n_pairs_at, n_pairs);
        k.patch32(n_sd_at, n_sd);
    }
    if (out_offsets) out_offsets[i1 - i0] = k.n;
    return (int64_t)k.n;
}

// The reference's stdout side effects for rows [i0, i1) (include/fg_hip.h).  LTSV: `println!("Missing value for name '{}'", name)`
// for every tab-separated part without ':' (ltsv_decoder.rs:93-101: `name` is the whole part), in order, up to the part where the
// decode stopped -- all of them for a row that decoded, the first hostname.off of them for a row that failed.
extern "C" int64_t fg_tables_stdout(fg_format fmt, fg_framing framing, const uint8_t* bytes, const uint64_t* offsets, const fg_tables* t,
                                    uint64_t i0, uint64_t i1, uint8_t* out, uint64_t cap) {
    if (!t || i1 < i0 || i1 > t->n || (i1 > i0 && (!offsets || !bytes || !t->meta))) return FG_ERR_ARG;
    if ((int)framing < 0 || (int)framing > 2) return FG_ERR_ARG;
    Sink k{out, cap};
    if (fmt != FG_LTSV) return 0;
    static const char kHead[] = "Missing value for name '";
    for (uint64_t i = i0; i < i1; ++i) {
        const uint32_t m = t->meta[i];
        if (!(FG_META_FLAGS(m) & FG_F_LTSV_NOVALUE)) continue;
        const uint8_t st = FG_META_STATUS(m);
        if (st == FG_ST_BAD_UTF8) continue;  // never decoded
        uint64_t b = offsets[i], e = offsets[i + 1];
        if (framing == FG_FRAME_LINE) {
            if (e > b && bytes[e - 1] == '\n') {
                --e;
                if (e > b && bytes[e - 1] == '\r') --e;
            }
        } else if (framing == FG_FRAME_NUL && e > b && bytes[e - 1] == 0) {
            --e;
        }
        uint64_t budget = (st == 0 || st == FG_ST_OVERFLOW) ? ~0ull : t->hostname ? (uint64_t)t->hostname[i].off : (uint64_t)FG_META_FACILITY(m);
        uint64_t ps = b;
        while (budget) {
            const uint8_t* tab = (const uint8_t*)memchr(bytes + ps, '\t', e - ps);
            const uint64_t pe = tab ? (uint64_t)(tab - bytes) : e;
            if (!memchr(bytes + ps, ':', pe - ps)) {
                k.put(kHead, sizeof(kHead) - 1);
                k.put(bytes + ps, pe - ps);
                k.put("'\n", 2);
                --budget;
            }
            if (!tab) break;
            ps = pe + 1;
        }
    }
    return (int64_t)k.n;
}
