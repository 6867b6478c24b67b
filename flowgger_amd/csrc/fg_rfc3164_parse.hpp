// fg_rfc3164_parse.hpp -- RFC3164Decoder::decode for ONE line, host + device.
// reference: src/flowgger/decoder/rfc3164_decoder.rs:31-213
//
//   decode            :31-50   parse_strip_pri, then the "standard" form, then the "custom" form; the custom form's
//                              error is the one that surfaces
//   decode_rfc_standard :57-88 [<pri>]<date> [tz] <hostname> <message...>, split on Unicode whitespace;
//                              msg = remaining tokens re-joined with ONE space (so: a span + FG_F_MSG_JOIN)
//   decode_rfc_custom :90-123  [<pri>]<hostname>: <date> [tz]: <message>  split on ": "
//   parse_strip_pri   :125-153
//   parse_date_token / parse_date :155-213   "[year] [month repr:short] [day padding:none] [hour]:[minute]:[second]",
//                              first with the CURRENT year prepended (configuration: the reference reads the clock),
//                              then with the year taken from the line; the token after the time may be an IANA zone
//                              name (time_tz::timezones::get_by_name -> the fg_tz_table of the configuration)
//
// One input can PANIC the reference (index out of bounds, :67: the date [+ zone] consumes every token of a line with
// more than three tokens): status ST_REF_PANIC.
#pragma once
#include <stdint.h>
#include <string.h>

#include "fg_timeconv.hpp"
#include "fg_unicode_ws.hpp"


namespace fg {
namespace r3164 {

enum : uint32_t {
    ST_OK = 0,
    ST_PRI_MALFORMED = 1,  // "Malformed RFC3164 event: Invalid priority"              :128-130
    ST_PRI_INVALID = 2,    // "Invalid priority"                                       :136
    ST_MALFORMED = 3,      // "Malformed RFC3164 event: Invalid timestamp or hostname"  :121
    ST_TIME_FORMAT = 4,    // "Invalid time format"                                    :158
    ST_DATE_YEAR = 5,      // "Unable to parse RFC3164 date with year"                 :176
    ST_DATE = 6,           // "Unable to parse the date in RFC3164 decoder"            :211
    ST_REF_PANIC = 7       // the reference panics (index out of bounds, :67)
};

// The zone table as the kernel sees it (built by fg_tz_index.hpp on the host).  A lookup is: two register-resident
// masks that reject most tokens without touching memory (first byte, length), one probe of an open-addressing hash
// table, one word-wise name comparison; the offset search starts from the span range that covers the configured year.
struct TzZone {
    uint32_t name_off, name_len;  // into name_words (byte offset, a multiple of 4)
    uint32_t first, last;         // the zone's spans [first, last] of utc_start / utc_off
    uint32_t y_lo, y_hi;          // the spans a local time inside [hint_lo, hint_hi) can fall into
};
struct TzView {
    const TzZone* zones;         // [nz], sorted by name (bytewise)
    uint32_t nz;
    const uint32_t* name_words;  // every name starts on a dword and is zero padded to the next one
    const uint32_t* slots;       // [slot_mask + 1]: (hash >> 16) << 16 | (zone + 1), 0 = empty; linear probing
    uint32_t slot_mask;
    const int64_t* utc_start;    // the offset utc_off[i] is in effect from utc_start[i] (first entry: INT64_MIN)
    const int32_t* utc_off;
    int64_t hint_lo, hint_hi;    // local seconds of the configured year [Jan 1, Jan 1 next year)
    uint64_t first_lo, first_hi; // bit b: some zone name starts with byte b (0..63 / 64..127); bytes >= 128 always pass
    uint64_t len_mask;           // bit min(len, 63): some zone name has this length
};
struct Cfg {
    int32_t current_year;
    TzView tz;
};
struct Row {
    uint32_t status = ST_OK;
    uint32_t fac = 0xFFu, sev = 0xFFu;  // 0xFF = None
    bool msg_join = false;                // msg = split_whitespace(span).join(" ")
    double ts = 0.0;
    uint32_t host_off = 0, host_len = 0, msg_off = 0, msg_len = 0, full_len = 0;
};

FG3_HD uint32_t ctz32(uint32_t v) { return (uint32_t)__builtin_ctz(v); }  // v != 0
// SWAR: nonzero iff some byte of w could be (part of) a White_Space character -- 0x20, anything below 0x0E (covers
// 0x09..0x0D), or a non-ASCII byte.  Conservative: a hit only means "look at the bytes".
FG3_HD uint32_t maybe_ws4(uint32_t w) {
    const uint32_t sp = w ^ 0x20202020u;
    return (w & 0x80808080u) | ((w - 0x0E0E0E0Eu) & ~w & 0x80808080u) | ((sp - 0x01010101u) & ~sp & 0x80808080u);
}
// str::split_whitespace: the next token of rd[pos .. end); false when there is none
template <class R>
FG3_HD bool next_token(R& rd, uint32_t& pos, uint32_t end, uint32_t* ts, uint32_t* te) {
    while (pos < end) {
        const uint32_t w = ws_at(rd, pos, end);
        if (!w) break;
        pos += w;
    }
    if (pos >= end) return false;
    *ts = pos;
    for (;;) {  // the token body: sixteen, then four bytes at a time while none of them can be whitespace
        // (a hostname is ONE memory round trip instead of six dependent ones: the lines in flight are bounded by LDS, so the
        //  length of this chain is the kernel's speed)
        while (pos + 16u <= end) {
            uint32_t q[4];
            rd.load16(pos, q);
            const uint32_t m0 = maybe_ws4(q[0]), m1 = maybe_ws4(q[1]), m2 = maybe_ws4(q[2]), m3 = maybe_ws4(q[3]);
            if (!(m0 | m1 | m2 | m3)) {
                pos += 16u;
                continue;
            }
            // the first byte that could be whitespace (flags sit in bit 7 of their byte; a borrow can only ADD flags above a hit)
            pos += m0 ? ctz32(m0) >> 3 : m1 ? 4u + (ctz32(m1) >> 3) : m2 ? 8u + (ctz32(m2) >> 3) : 12u + (ctz32(m3) >> 3);
            break;
        }
        while (pos + 4u <= end && !maybe_ws4(rd.load4(pos, 4u))) pos += 4u;
        if (pos >= end || ws_at(rd, pos, end)) break;  // (continuation bytes are never whitespace lead bytes)
        ++pos;
    }
    *te = pos;
    return true;
}
// ---- tokens out of a 64-byte REGISTER window ------------------------------------------------------------------------
// The date, the zone and the hostname are four to six short tokens at the start of the line: read one by one, each costs a
// dependent memory round trip (the next token starts where the last one ended).  The window holds, for 64 bytes from `w0`, one
// bit per byte for "is an ASCII space" (exact) and one for "could be some OTHER White_Space character" (conservative: a tab, a
// line feed, any non-ASCII byte ...): as long as no `other` bit lies inside what a token request looks at, the split is decided
// from the two masks; anything else -- and everything beyond the window -- is next_token's business.
struct TokWin {
    uint32_t w0 = 0;
    uint64_t sp = 0, other = 0;
    bool valid = false;
};
FG3_HD uint32_t gather4(uint32_t flags) {  // flags in bit 7 of each byte -> four bits
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_udot4(flags >> 7, 0x08040201u, 0u, false);
#else
    return (((flags >> 7) * 0x00204081u) >> 21) & 0xFu;
#endif
}
template <class R>
FG3_HD void tokwin_load(R& rd, TokWin& w, uint32_t pos, uint32_t end, uint32_t* head0 = nullptr, uint32_t* head1 = nullptr) {
    w.valid = pos + 64u <= end;  // (sixty-four WANTED bytes: the readers do not read beyond them)
    if (!w.valid) return;
    w.w0 = pos;
    uint32_t q[16];
    rd.load16(pos, q);
    rd.load16(pos + 16u, q + 4);
    rd.load16(pos + 32u, q + 8);
    rd.load16(pos + 48u, q + 12);
    uint64_t sp = 0, maybe = 0;
    for (int k = 0; k < 16; ++k) {
        const uint32_t x = q[k] ^ 0x20202020u;
        const uint32_t is_sp = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;  // exact
        sp |= (uint64_t)gather4(is_sp) << (4 * k);
        maybe |= (uint64_t)gather4(maybe_ws4(q[k])) << (4 * k);
    }
    w.sp = sp;
    w.other = maybe & ~sp;
    if (head0) {  // the window's first eight bytes (the line's "<PRI>")
        *head0 = q[0];
        *head1 = q[1];
    }
}
// next_token through the window (loaded at the first request, kept while requests stay inside it)
template <class R>
FG3_HD bool next_token_w(R& rd, TokWin& w, uint32_t& pos, uint32_t end, uint32_t* ts, uint32_t* te) {
    if (!w.valid || pos - w.w0 >= 64u) {
        if (w.valid || pos == w.w0) tokwin_load(rd, w, pos, end);  // (a window that could not be loaded once is not retried further in)
    }
    if (w.valid && pos - w.w0 < 64u) {
        const uint32_t off = pos - w.w0;
        const uint64_t nonsp = ~w.sp >> off;
        if (nonsp != 0ull) {
            const uint32_t a = off + (uint32_t)__builtin_ctzll(nonsp);
            const uint64_t sp2 = w.sp >> a;
            if (sp2 != 0ull) {
                const uint32_t b = a + (uint32_t)__builtin_ctzll(sp2);  // the space that ends the token: a < b <= 63
                const uint64_t span = (b - off >= 63u ? ~0ull : ((1ull << (b - off + 1u)) - 1ull)) << off;
                if ((w.other & span) == 0ull) {
                    *ts = w.w0 + a;
                    *te = w.w0 + b;
                    pos = w.w0 + b;
                    return true;
                }
            }
        }
    }
    return next_token(rd, pos, end, ts, te);
}

// str::trim_end of rd[s .. e): the new end
template <class R>
FG3_HD uint32_t trim_end_ws(R& rd, uint32_t s, uint32_t e) {
    while (e > s) {
        const uint32_t c = rd.byte(e - 1);
        if (c < 0x80u) {
            if (!(c == 32u || (c - 9u) <= 4u)) break;
            e -= 1;
        } else if (e - s >= 2 && ws_at(rd, e - 2, e) == 2u) {
            e -= 2;
        } else if (e - s >= 3 && ws_at(rd, e - 3, e) == 3u) {
            e -= 3;
        } else {
            break;
        }
    }
    return e;
}

struct Tok {
    uint32_t s, e;
};

template <class R>
FG3_HD bool two_digits(R& rd, uint32_t at, uint32_t* v) {
    const uint32_t a = rd.byte(at) - '0', b = rd.byte(at + 1) - '0';
    if (a > 9u || b > 9u) return false;
    *v = a * 10u + b;
    return true;
}
// [month repr:short] [day padding:none] [hour]:[minute]:[second] from three whole tokens + a year -> seconds of the
// PrimitiveDateTime taken as UTC
template <class R>
FG3_HD bool parse_mdt(R& rd, const Tok& mon, const Tok& day, const Tok& tim, int year, int64_t* local_secs) {
    const uint32_t dl = day.e - day.s;
    if (mon.e - mon.s != 3u || dl < 1u || dl > 2u || tim.e - tim.s != 8u) return false;
    // the three tokens' bytes with every load in flight (byte by byte, with a verdict after each, they were a chain of dependent
    // memory round trips), then everything out of registers
    const uint32_t mw = rd.load4(mon.s, 3u) & 0xFFFFFFu, dw = rd.load4(day.s, dl), t0 = rd.load4(tim.s, 4u), t1 = rd.load4(tim.s + 4u, 4u);
    int month = 0;
    switch (mw) {  // case-sensitive; little endian: first letter in the low byte
        case 0x6E614Au: month = 1; break;   // Jan
        case 0x626546u: month = 2; break;   // Feb
        case 0x72614Du: month = 3; break;   // Mar
        case 0x727041u: month = 4; break;   // Apr
        case 0x79614Du: month = 5; break;   // May
        case 0x6E754Au: month = 6; break;   // Jun
        case 0x6C754Au: month = 7; break;   // Jul
        case 0x677541u: month = 8; break;   // Aug
        case 0x706553u: month = 9; break;   // Sep
        case 0x74634Fu: month = 10; break;  // Oct
        case 0x766F4Eu: month = 11; break;  // Nov
        case 0x636544u: month = 12; break;  // Dec
        default: return false;
    }
    uint32_t d = (dw & 0xFFu) - '0';
    if (d > 9u) return false;
    if (dl == 2u) {
        const uint32_t d2 = ((dw >> 8) & 0xFFu) - '0';
        if (d2 > 9u) return false;
        d = d * 10u + d2;
    }
    // "HH:M" "M:SS"
    const uint32_t h1 = (t0 & 0xFFu) - '0', h2 = ((t0 >> 8) & 0xFFu) - '0', m1 = (t0 >> 24) - '0';
    const uint32_t m2 = (t1 & 0xFFu) - '0', s1 = ((t1 >> 16) & 0xFFu) - '0', s2 = (t1 >> 24) - '0';
    if (h1 > 9u || h2 > 9u || m1 > 9u || m2 > 9u || s1 > 9u || s2 > 9u || ((t0 >> 16) & 0xFFu) != ':' || ((t1 >> 8) & 0xFFu) != ':') return false;
    const uint32_t hh = h1 * 10u + h2, mm = m1 * 10u + m2, ss = s1 * 10u + s2;
    if (hh > 23u || mm > 59u || ss > 59u) return false;
    if (d < 1u || (int)d > days_in_month(year, month)) return false;
    *local_secs = days_from_civil(year, month, (int)d) * 86400ll + (int64_t)(hh * 3600u + mm * 60u + ss);
    return true;
}
// [year]: optional sign, exactly four digits (time 0.3, no large-dates)
template <class R>
FG3_HD bool parse_year_tok(R& rd, const Tok& t, int* year) {
    uint32_t s = t.s;
    bool neg = false;
    if (s < t.e) {
        const uint32_t c = rd.byte(s);
        if (c == '-' || c == '+') {
            neg = c == '-';
            ++s;
        }
    }
    if (t.e - s != 4u) return false;
    int v = 0;
    for (uint32_t k = 0; k < 4u; ++k) {
        const uint32_t d = rd.byte(s + k) - '0';
        if (d > 9u) return false;
        v = v * 10 + (int)d;
    }
    *year = neg ? -v : v;
    return true;
}
// hash of a zone name / a token (FNV-1a, 32 bit)
FG3_HD uint32_t tz_hash_step(uint32_t h, uint32_t byte) { return (h ^ byte) * 16777619u; }
constexpr uint32_t kTzHashInit = 2166136261u;

// time_tz::timezones::get_by_name: exact, case-sensitive match of the whole token; -1 = not a zone name
template <class R>
FG3_HD int32_t tz_lookup(R& rd, const Tok& t, const TzView& tz) {
    const uint32_t len = t.e - t.s;
    const uint32_t c0 = rd.byte(t.s);
    const uint64_t fm = c0 < 64u ? tz.first_lo : tz.first_hi;
    if (c0 < 128u && !((fm >> (c0 & 63u)) & 1ull)) return -1;
    if (!((tz.len_mask >> (len < 63u ? len : 63u)) & 1ull)) return -1;
    uint32_t h = kTzHashInit;
    for (uint32_t k = 0; k < len; ++k) h = tz_hash_step(h, rd.byte(t.s + k));
    for (uint32_t slot = h & tz.slot_mask;; slot = (slot + 1u) & tz.slot_mask) {
        const uint32_t v = tz.slots[slot];
        if (v == 0u) return -1;
        if ((v >> 16) != (h >> 16)) continue;
        const uint32_t z = (v & 0xFFFFu) - 1u;
        const TzZone zr = tz.zones[z];
        if (zr.name_len != len) continue;
        const uint32_t* nw = tz.name_words + (zr.name_off >> 2);
        uint32_t diff = 0;
        for (uint32_t k = 0; k < len; k += 4u) {
            const uint32_t w = nw[k >> 2];
            const uint32_t nb = len - k < 4u ? len - k : 4u;
            for (uint32_t j = 0; j < nb; ++j) diff |= rd.byte(t.s + k + j) ^ ((w >> (8u * j)) & 0xFFu);
        }
        if (diff == 0u) return (int32_t)z;
    }
}
// PrimitiveDateTime::assume_timezone: the UTC offset for a LOCAL time: the first span (chronologically) whose local
// end lies after it -- the earlier offset for an ambiguous time, the later one inside a gap (UNPINNED)
FG3_HD int32_t tz_offset_local(const TzView& tz, uint32_t zone, int64_t local) {
    const TzZone zr = tz.zones[zone];
    const bool hinted = local >= tz.hint_lo && local < tz.hint_hi;
    uint32_t lo = hinted ? zr.y_lo : zr.first, hi = hinted ? zr.y_hi : zr.last;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (local < tz.utc_start[mid + 1] + (int64_t)tz.utc_off[mid]) hi = mid;
        else lo = mid + 1;
    }
    return tz.utc_off[lo];
}

// parse_date_token (:155-162) over the tokens of rd[pos .. end), read one at a time (no token array: a dynamically
// indexed array would live in scratch memory on the GPU).  On ST_OK: *nx = the first token after the date [+ zone]
// (*have_nx = false: there is none) and pos stands behind it.
template <class R>
FG3_HD uint32_t parse_date_token(R& rd, uint32_t& pos, uint32_t end, const Cfg& cfg, bool need4, double* ts, Tok* nx, bool* have_nx,
                                 TokWin& win) {
    Tok t0, t1, t2, t3;
    if (!next_token_w(rd, win, pos, end, &t0.s, &t0.e) || !next_token_w(rd, win, pos, end, &t1.s, &t1.e) ||
        !next_token_w(rd, win, pos, end, &t2.s, &t2.e))
        return ST_TIME_FORMAT;  // fewer than three tokens
    const bool have3 = next_token_w(rd, win, pos, end, &t3.s, &t3.e);
    if (need4 && !have3) return ST_TIME_FORMAT;  // the standard form is only tried on more than three tokens (:61-63)
    int64_t local = 0;
    Tok cand = t3;
    bool have = have3;
    if (!parse_mdt(rd, t0, t1, t2, cfg.current_year, &local)) {
        if (!have3) return ST_DATE_YEAR;
        int year;
        if (!parse_year_tok(rd, t0, &year) || !parse_mdt(rd, t1, t2, t3, year, &local)) return ST_DATE;
        have = next_token_w(rd, win, pos, end, &cand.s, &cand.e);
    }
    int64_t off = 0;
    if (have && cfg.tz.nz) {
        const int32_t z = tz_lookup(rd, cand, cfg.tz);
        if (z >= 0) {
            off = tz_offset_local(cfg.tz, (uint32_t)z, local);
            have = next_token_w(rd, win, pos, end, &cand.s, &cand.e);
        }
    }
    *ts = unix_nanos_to_f64(local - off, 0u);
    *nx = cand;
    *have_nx = have;
    return ST_OK;
}

// the whole decoder for rd[0 .. len)
template <class R>
FG3_HD void parse_line(R& rd, uint32_t len, const Cfg& cfg, Row& r) {
    r = Row{};
    // parse_strip_pri.  The line's first 64 bytes go into the register window the date tokens are read from (ONE memory round
    // trip); "<PRI>" is decided from its first eight bytes when its '>' lies there -- else, and for short lines, byte by byte.
    uint32_t q0 = 0;
    TokWin lwin;
    uint32_t h0 = 0, h1 = 0;
    tokwin_load(rd, lwin, 0u, len, &h0, &h1);
    bool pri_done = false;
    if (lwin.valid) {
        if ((h0 & 0xFFu) != '<') {
            pri_done = true;  // no priority
        } else {
            const uint64_t hv = (uint64_t)h0 | ((uint64_t)h1 << 32);
            uint32_t gt = 0;
            for (uint32_t i = 7; i >= 1u; --i)
                if (((hv >> (8u * i)) & 0xFFu) == '>') gt = i;  // the first '>' of bytes 1..7
            if (gt != 0u) {
                auto hb = [&](uint32_t i) -> uint32_t { return (uint32_t)(hv >> (8u * i)) & 0xFFu; };  // i <= gt <= 7
                uint32_t a = 0, b = gt + 1;
                while (a < b && hb(a) == '<') ++a;
                while (b > a && hb(b - 1) == '>') --b;
                if (a < b && hb(a) == '+') ++a;
                if (a >= b) {
                    r.status = ST_PRI_INVALID;
                    return;
                }
                uint32_t v = 0;
                for (uint32_t i = a; i < b; ++i) {
                    const uint32_t d = hb(i) - '0';
                    if (d > 9u || (v = v * 10u + d) > 255u) {
                        r.status = ST_PRI_INVALID;
                        return;
                    }
                }
                r.fac = v >> 3;
                r.sev = v & 7u;
                q0 = gt + 1;
                pri_done = true;
            }
        }
    }
    if (!pri_done && len && rd.byte(0) == '<') {
        uint32_t gt = 0;
        bool found = false;
        for (uint32_t i = 1; i < len; ++i)
            if (rd.byte(i) == '>') {
                gt = i;
                found = true;
                break;
            }
        if (!found) {
            r.status = ST_PRI_MALFORMED;
            return;
        }
        // "<...>".trim_start_matches('<').trim_end_matches('>') parsed as u8: [+]digits, <= 255
        uint32_t a = 0, b = gt + 1;
        while (a < b && rd.byte(a) == '<') ++a;
        while (b > a && rd.byte(b - 1) == '>') --b;
        if (a < b && rd.byte(a) == '+') ++a;
        if (a >= b) {
            r.status = ST_PRI_INVALID;
            return;
        }
        uint32_t v = 0;
        for (uint32_t i = a; i < b; ++i) {
            const uint32_t d = rd.byte(i) - '0';
            if (d > 9u || (v = v * 10u + d) > 255u) {
                r.status = ST_PRI_INVALID;
                return;
            }
        }
        r.fac = v >> 3;
        r.sev = v & 7u;
        q0 = gt + 1;
    }
    r.full_len = trim_end_ws(rd, 0, len);  // line.trim_end()

    // ---- decode_rfc_standard ---------------------------------------------------------------------------------------
    {   // needs more than three tokens (:61-63); then the date [+ zone], the hostname, the message tokens
        uint32_t pos = q0;
        {
            double ts;
            Tok host;
            bool have_host;
            if (parse_date_token(rd, pos, len, cfg, true, &ts, &host, &have_host, lwin) == ST_OK) {
                if (!have_host) {  // the date [+ zone] consumed every token: index out of bounds in the reference (:67)
                    r.status = ST_REF_PANIC;
                    return;
                }
                r.ts = ts;
                r.host_off = host.s;
                r.host_len = host.e - host.s;
                Tok m;
                if (next_token(rd, pos, len, &m.s, &m.e)) {
                    r.msg_off = m.s;
                    r.msg_len = trim_end_ws(rd, r.msg_off, len) - r.msg_off;
                    r.msg_join = true;
                } else {
                    r.msg_off = host.e;
                    r.msg_len = 0;
                }
                return;
            }
        }
    }
    // ---- decode_rfc_custom -----------------------------------------------------------------------------------------
    uint32_t p1 = 0, p2 = 0, found = 0;
    for (uint32_t i = q0; i + 1u < len;) {  // msg.split(": "): non-overlapping, left to right
        if (i + 16u <= len) {  // sixteen bytes without a ':' cannot start a separator: skip to the first ':' (or past them)
            uint32_t q[4];
            rd.load16(i, q);
            uint32_t h[4];
            for (int k = 0; k < 4; ++k) {
                const uint32_t c4 = q[k] ^ 0x3A3A3A3Au;
                h[k] = (c4 - 0x01010101u) & ~c4 & 0x80808080u;
            }
            if (!(h[0] | h[1] | h[2] | h[3])) {
                i += 16u;
                continue;
            }
            // (the lowest flag is exact: a borrow only adds flags above a hit)
            i += h[0] ? ctz32(h[0]) >> 3 : h[1] ? 4u + (ctz32(h[1]) >> 3) : h[2] ? 8u + (ctz32(h[2]) >> 3) : 12u + (ctz32(h[3]) >> 3);
            if (i + 1u >= len) break;  // (a ':' as the last byte)
        } else if (i + 4u <= len) {  // four bytes without a ':' cannot start a separator
            const uint32_t c4 = rd.load4(i, 4u) ^ 0x3A3A3A3Au;
            if (!((c4 - 0x01010101u) & ~c4 & 0x80808080u)) {
                i += 4u;
                continue;
            }
        }
        if (rd.byte(i) == ':' && rd.byte(i + 1) == ' ') {
            if (found == 0) p1 = i;
            else p2 = i;
            if (++found == 2u) break;
            i += 2;
        } else {
            ++i;
        }
    }
    if (found < 2u) {
        r.status = ST_MALFORMED;
        return;
    }
    uint32_t pos = p1 + 2u;
    double ts;
    Tok nx;
    bool have_nx;
    TokWin cwin;
    cwin.w0 = pos;
    const uint32_t st = parse_date_token(rd, pos, p2, cfg, false, &ts, &nx, &have_nx, cwin);
    if (st != ST_OK) {
        r.status = st;
        return;
    }
    r.ts = ts;
    r.host_off = q0;
    r.host_len = p1 - q0;
    r.msg_off = p2 + 2u;
    r.msg_len = len - r.msg_off;
}

}  // namespace r3164
}  // namespace fg
