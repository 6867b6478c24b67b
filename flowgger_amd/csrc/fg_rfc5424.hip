// fg_rfc5424.hip -- gfx950 kernel for RFC5424Decoder::decode
// (reference: src/flowgger/decoder/rfc5424_decoder.rs:17-242).
//
// HBM-bound byte/integer work, no MFMA.  One 64-lane wavefront owns 64 consecutive lines = one
// contiguous byte range of the packed buffer, and works in two stages:
//
//   stage A (byte-parallel, coalesced): the range is streamed HBM -> VGPR with 16 B per lane
//     (1 KiB per wave-instruction, every byte fetched once).  While the data sits in registers
//     each lane derives a 16-bit "is 0x20" mask for its 16 bytes with SWAR arithmetic; data and
//     mask go to the wave's LDS tile (ds_write_b128 + ds_write_b16, both conflict-free).
//   stage B (lane-per-line): every lane tokenises ITS line.  Field boundaries (splitn(7,' '))
//     come from a 128-bit window of the space bitmap with ctz / clear-lowest-bit; <PRI>1 and the
//     fixed-format RFC3339 timestamp are parsed from a handful of unaligned dword reads with
//     SWAR digit validation; trims look at the last / first bytes only.  No per-byte loops on
//     the common path, so the 64 lanes stay convergent.
//   Anything unusual (BOM, '+' in the priority, >=16 fraction digits, header beyond 128 bytes,
//   multi-byte whitespace at the trim points, structured data) drops into the generic
//   byte-walking parser below, which is also what a line too long for the LDS tile uses,
//   reading straight from global memory.
//
// Fixed-width results go to struct-of-array tables (one coalesced store per column);
// structured-data entries get their slots from ONE wave-aggregated atomic (count pass -> wave
// prefix sum -> fill pass, both out of LDS).
#include <cstdlib>

#include "fg_device.hpp"

namespace fg {

// status codes == index into the reference's error strings (fg_error_string)
enum : uint32_t {
    E_OK = 0,
    E_BOM = 1,        // "Unsupported BOM"                               :69
    E_BRACKETS = 2,   // "The priority should be inside brackets"        :76
    E_INVPRI = 3,     // "Invalid priority"                              :83
    E_NOVER = 4,      // "Missing version"                               :84
    E_BADVER = 5,     // "Unsupported version"                           :86
    E_NOTS = 6,       // "Missing timestamp"                             :25
    E_BADTS = 7,      // "Unable to parse the date from RFC3339 ..."     :97
    E_NOHOST = 8,     // "Missing hostname"                              :26
    E_NOAPP = 9,      // "Missing application name"                      :27
    E_NOPROC = 10,    // "Missing process id"                            :28
    E_NOMSGID = 11,   // "Missing message id"                            :29
    E_NODATA = 12,    // "Missing message data"                          :30
    E_NOMSG = 13,     // "Missing log message"                           :129,:148
    E_MALFORMED = 14, // "Malformated RFC5424 message"                   :154,:159
    E_NOSD = 15,      // "Missing structured data"                       :177
    E_SDFMT = 16,     // "Format error in the structured data"           :235
    E_NOBRACKET = 17  // "Missing ] after structured data"               :239
};

struct Row {
    uint32_t status = E_OK;
    uint32_t facility = 0xFF, severity = 0xFF, flags = 0;
    double ts = 0.0;
    uint32_t off[6];
    uint32_t len[6];
    uint32_t data0 = 0;  // index of part 7 ("[..." / "-...") for the fill pass
    uint32_t n_ent = 0;
};

// =============================================================================================
// Generic byte-walking parser (any reader).  Restates decode() statement by statement.
// =============================================================================================

template <class R>
__device__ __forceinline__ uint32_t find_space(R& rd, uint32_t q, uint32_t len) {
    while (q < len && rd.byte(q) != ' ') ++q;
    return q;
}

// parse_data's structured-data walk (rfc5424_decoder.rs:134-158 + parse_sd_data :174-242).
// pos = index of the first '['.  EMIT=false counts entries; EMIT=true writes them starting at
// slot `slot`.  On success *msg_at = index of the ' ' that starts the message.
template <bool EMIT, class R>
__device__ __forceinline__ uint32_t sd_walk(R& rd, uint32_t pos, uint32_t len, uint32_t* msg_at, uint32_t* n_ent,
                            const DevTables& t, uint32_t slot) {
    uint32_t cnt = 0;
    for (;;) {
        // sd_id = bytes after '[' up to the first ' ' (anything allowed)            :175-177
        uint32_t s = pos + 1;
        uint32_t sp = find_space(rd, s, len);
        if (sp >= len) return E_NOSD;
        if (EMIT) {
            t.ent_name[slot + cnt] = fg_span{s, sp - s};
            t.ent_val[slot + cnt] = 0;
            t.ent_type[slot + cnt] = FG_T_SDID;
            t.ent_flags[slot + cnt] = 0;
        }
        ++cnt;
        // 5-state machine equivalent to the reference's 6-tuple match             :187-237
        //   0 OUT  1 IN_NAME  2 HAVE_NAME (expects '"')  3 IN_VALUE  4 ESC
        uint32_t st = 0, name_s = 0, name_e = 0, val_s = 0, esc_seen = 0;
        uint32_t i = sp + 1;
        uint32_t after = 0;
        for (; i < len; ++i) {
            uint32_t c = rd.byte(i);
            if (st == 3) {
                if (c == '\\') {
                    st = 4;
                    esc_seen = 1;
                } else if (c == '"') {
                    if (EMIT) {
                        t.ent_name[slot + cnt] = fg_span{name_s, name_e - name_s};
                        t.ent_val[slot + cnt] = (uint64_t)val_s | ((uint64_t)(i - val_s) << 32);
                        t.ent_type[slot + cnt] = FG_T_STRING;
                        t.ent_flags[slot + cnt] = esc_seen ? FG_EF_VAL_ESC : 0;
                    }
                    ++cnt;
                    st = 0;
                }
            } else if (st == 4) {
                st = 3;
            } else {
                bool is_name = (c - 33u) <= 93u && c != '"' && c != '=' && c != ']';  // :188-192
                if (st == 0) {
                    if (c == ' ' || c == '"') {
                        // contextless space / tolerated stray quote                 :194,:232
                    } else if (c == ']') {
                        after = i + 1;  //                                              :197
                        break;
                    } else if (is_name) {
                        st = 1;
                        name_s = i;
                    } else {
                        return E_SDFMT;
                    }
                } else if (st == 1) {
                    if (is_name) {
                    } else if (c == '=') {
                        name_e = i;
                        st = 2;
                    } else {
                        return E_SDFMT;
                    }
                } else {  // st == 2
                    if (c != '"') return E_SDFMT;
                    st = 3;
                    val_s = i + 1;
                    esc_seen = 0;
                }
            }
        }
        if (after == 0) return E_NOBRACKET;   // :239
        if (after >= len) return E_NOMSG;     // :148
        uint32_t c = rd.byte(after);
        if (c == '[') {
            pos = after;
            continue;
        }
        if (c != ' ') return E_MALFORMED;     // :154
        *msg_at = after;
        *n_ent = cnt;
        return E_OK;
    }
}

// parse_data (:127-161) + parse_msg (:163-172) + full_msg (:46) for part 7 = [q, len).
template <class R>
__device__ __forceinline__ void parse_tail(R& rd, uint32_t q, uint32_t line0, uint32_t len, Row& r, const DevTables& t) {
    if (q >= len) {
        r.status = E_NOMSG;
        return;
    }
    r.data0 = q;
    uint32_t c = rd.byte(q);
    uint32_t msg_at;
    if (c == '-') {
        msg_at = q + 1;
    } else if (c == '[') {
        uint32_t st = sd_walk<false>(rd, q, len, &msg_at, &r.n_ent, t, 0);
        if (st != E_OK) {
            r.status = st;
            r.n_ent = 0;
            return;
        }
    } else {
        r.status = E_MALFORMED;
        return;
    }
    // full_msg = line.trim_end(); msg = rest.trim() -- both end at the last non-whitespace
    // char of the line, so one backward scan serves both.
    uint32_t e = trim_end(rd, line0, len);
    r.off[S_FULL] = line0;
    r.len[S_FULL] = e - line0;
    uint32_t s = trim_start(rd, msg_at, len);
    if (e > s) {
        r.off[S_MSG] = s;
        r.len[S_MSG] = e - s;
    }
}

// Everything of decode() except writing SD entries.
template <class R>
__device__ __forceinline__ void parse_line_generic(R& rd, uint32_t len, Row& r, const DevTables& t) {
    uint32_t p = 0;
    // BOM::parse :62-72
    if (len >= 3 && rd.byte(0) == 0xEFu && rd.byte(1) == 0xBBu && rd.byte(2) == 0xBFu) {
        p = 3;
        r.flags |= FG_F_BOM;
    } else if (len == 0 || rd.byte(0) != '<') {
        r.status = E_BOM;
        return;
    }
    const uint32_t line0 = p;
    // parse_pri_version :74-92 over part 1 = [p, first ' ')
    if (p >= len || rd.byte(p) != '<') {
        r.status = E_BRACKETS;
        return;
    }
    uint32_t q = p + 1;
    {
        // u8::from_str of the text before the first '>' (or the whole part)
        uint32_t v = 0, nd = 0;
        bool ok = true;
        if (q < len && rd.byte(q) == '+') ++q;
        while (q < len) {
            uint32_t c = rd.byte(q);
            if (c == '>' || c == ' ') break;
            uint32_t d = c - '0';
            if (d <= 9u) {
                v = v * 10u + d;
                if (v > 255u) {
                    v = 256u;
                    ok = false;
                }
                ++nd;
            } else {
                ok = false;
            }
            ++q;
        }
        if (!ok || nd == 0) {
            r.status = E_INVPRI;
            return;
        }
        if (q >= len || rd.byte(q) != '>') {
            r.status = E_NOVER;
            return;
        }
        ++q;
        // version must be exactly "1" up to the end of the part
        if (!(q < len && rd.byte(q) == '1' && (q + 1 == len || rd.byte(q + 1) == ' '))) {
            r.status = E_BADVER;
            return;
        }
        ++q;
        r.facility = v >> 3;
        r.severity = v & 7u;
    }
    if (q >= len) {
        r.status = E_NOTS;
        return;
    }
    ++q;  // the ' ' after part 1
    {
        uint32_t e = find_space(rd, q, len);
        if (!parse_rfc3339(rd, q, e, &r.ts)) {
            r.status = E_BADTS;
            return;
        }
        q = e;
    }
    // hostname, appname, procid, msgid: verbatim parts :26-29
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (q >= len) {
            r.status = E_NOHOST + k;
            return;
        }
        ++q;
        uint32_t e = find_space(rd, q, len);
        r.off[k] = q;
        r.len[k] = e - q;
        q = e;
    }
    if (q >= len) {
        r.status = E_NODATA;
        return;
    }
    ++q;
    parse_tail(rd, q, line0, len, r, t);
}

// =============================================================================================
// Stage A: SWAR byte classification
// =============================================================================================

// 4-bit mask (bit i = byte i of x equals 0x20), exact (no borrow artefacts).
__device__ __forceinline__ uint32_t space_nibble(uint32_t x) {
    uint32_t y = x ^ 0x20202020u;                                          // zero byte <=> space
    uint32_t t = ((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y;                    // bit7 set <=> byte != 0
    t = ~t & 0x80808080u;                                                  // bit7 set <=> byte == 0
    return ((t >> 7) * 0x00204081u) >> 21 & 0xFu;                          // gather bits 0,8,16,24
}
__device__ __forceinline__ uint32_t space_mask16(const uint4& v) {
    return space_nibble(v.x) | (space_nibble(v.y) << 4) | (space_nibble(v.z) << 8) | (space_nibble(v.w) << 12);
}

// =============================================================================================
// Stage B: register-resident fast path
// =============================================================================================

struct Tile {
    const uint32_t* w;   // tile bytes as dwords (LDS)
    const uint32_t* bm;  // space bitmap, bit i <=> tile byte i == ' ' (LDS)
};

// 8 consecutive bytes starting at tile byte `a` (unaligned) as two little-endian dwords
__device__ __forceinline__ void load8(const Tile& T, uint32_t a, uint32_t* lo, uint32_t* hi) {
    uint32_t d = a >> 2, s = a & 3u;
    uint32_t w0 = T.w[d], w1 = T.w[d + 1], w2 = T.w[d + 2];
    *lo = __builtin_amdgcn_alignbyte(w1, w0, s);
    *hi = __builtin_amdgcn_alignbyte(w2, w1, s);
}
// first space at line index >= q (tile byte base+q), or len.  Bitmap walk, 32 bytes per step.
__device__ __forceinline__ uint32_t find_space_bm(const Tile& T, uint32_t base, uint32_t q, uint32_t len) {
    while (q < len) {
        uint32_t a = base + q;
        uint32_t w = T.bm[a >> 5] >> (a & 31u);
        if (w) {
            uint32_t r = q + (uint32_t)__builtin_ctz(w);
            return r < len ? r : len;
        }
        q += 32u - (a & 31u);
    }
    return len;
}
// every byte of x (already XORed with the expected pattern) must be <= its limit, where
// add = 0x7F - limit per byte: returns nonzero iff some byte exceeds its limit.
__device__ __forceinline__ uint32_t swar_exceeds(uint32_t x, uint32_t add) {
    return (x | ((x & 0x7F7F7F7Fu) + add)) & 0x80808080u;
}
// four ASCII-digit values (0..9 per byte, first digit in the low byte) -> 0..9999
__device__ __forceinline__ uint32_t digits4(uint32_t x) {
    uint32_t t = (x * 10u + (x >> 8)) & 0x00FF00FFu;  // byte0 = 10*d0+d1, byte2 = 10*d2+d3
    return (t & 0xFFu) * 100u + (t >> 16);
}

// RFC3339 from registers.  [t0, t0+L) is the timestamp part (line-relative).  Returns
// 1 = ok (*out set), 0 = invalid, 2 = undecided (caller falls back to the byte-wise parser).
__device__ __forceinline__ int fast_rfc3339(const Tile& T, uint32_t base, uint32_t t0, uint32_t L, double* out) {
    if (L < 20u) return 0;
    const uint32_t a = base + t0;
    const uint32_t d = a >> 2, s = a & 3u;
    uint32_t w[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) w[k] = T.w[d + k];
    uint32_t r[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) r[k] = __builtin_amdgcn_alignbyte(w[k + 1], w[k], s);
    // bytes 0..18 = "YYYY-MM-DDtHH:MM:SS" ; XOR with the pattern: digits -> 0..9, literals -> 0
    uint32_t x0 = r[0] ^ 0x30303030u;                         // Y Y Y Y
    uint32_t x1 = r[1] ^ 0x2D30302Du;                         // - M M -
    uint32_t x2 = (r[2] | 0x00200000u) ^ 0x30743030u;         // D D t H   ('T'|0x20 == 't')
    uint32_t x3 = r[3] ^ 0x30303A30u;                         // H : M M
    uint32_t x4 = (r[4] & 0x00FFFFFFu) ^ 0x0030303Au;         // : S S (byte 19 cleared)
    uint32_t bad = swar_exceeds(x0, 0x76767676u) | swar_exceeds(x1, 0x7F76767Fu) | swar_exceeds(x2, 0x767F7676u) |
                   swar_exceeds(x3, 0x76767F76u) | swar_exceeds(x4, 0x7F76767Fu);
    if (bad) return 0;
    DateTimeParts p;
    p.year = (int)digits4(x0);
    p.month = (int)(((x1 >> 8) & 0xFFu) * 10u + ((x1 >> 16) & 0xFFu));
    p.day = (int)((x2 & 0xFFu) * 10u + ((x2 >> 8) & 0xFFu));
    p.hour = (int)((x2 >> 24) * 10u + (x3 & 0xFFu));
    p.minute = (int)(((x3 >> 16) & 0xFFu) * 10u + (x3 >> 24));
    p.second = (int)(((x4 >> 8) & 0xFFu) * 10u + ((x4 >> 16) & 0xFFu));
    p.nano = 0;
    uint32_t pos = 19;  // index of the time-zone designator
    if ((r[4] >> 24) == '.') {
        // fraction digits live in bytes 20..35 = r[5..8]; count the leading digits (clamped to L)
        uint32_t f0 = r[5] ^ 0x30303030u, f1 = r[6] ^ 0x30303030u, f2 = r[7] ^ 0x30303030u, f3 = r[8] ^ 0x30303030u;
        uint32_t n0 = swar_exceeds(f0, 0x76767676u), n1 = swar_exceeds(f1, 0x76767676u);
        uint32_t n2 = swar_exceeds(f2, 0x76767676u), n3 = swar_exceeds(f3, 0x76767676u);
        uint32_t nd = n0 ? (uint32_t)__builtin_ctz(n0) >> 3
                         : n1 ? 4u + ((uint32_t)__builtin_ctz(n1) >> 3)
                              : n2 ? 8u + ((uint32_t)__builtin_ctz(n2) >> 3)
                                   : n3 ? 12u + ((uint32_t)__builtin_ctz(n3) >> 3) : 16u;
        uint32_t room = L - 20u;
        if (nd > room) nd = room;
        if (nd == 0) return 0;
        if (nd >= 16u) return 2;  // very long fraction: let the byte-wise parser decide
        // keep the first min(nd,9) digits, zero the rest => nine digits with trailing zeros
        uint32_t keep = nd < 9u ? nd : 9u;
        uint32_t k0 = keep >= 4u ? 0xFFFFFFFFu : (1u << (8u * keep)) - 1u;
        uint32_t k1 = keep >= 8u ? 0xFFFFFFFFu : keep <= 4u ? 0u : (1u << (8u * (keep - 4u))) - 1u;
        uint32_t d8 = keep >= 9u ? (f2 & 0xFFu) : 0u;
        p.nano = (digits4(f0 & k0) * 10000u + digits4(f1 & k1)) * 10u + d8;
        pos = 20u + nd;
    }
    if (pos >= L) return 0;
    // time-zone designator: re-read 8 bytes at its (data-dependent) position
    uint32_t z0, z1;
    load8(T, a + pos, &z0, &z1);
    uint32_t c = z0 & 0xFFu;
    p.off_sign = 1;
    p.off_h = 0;
    p.off_m = 0;
    if ((c | 0x20u) == 'z') {
        if (pos + 1u != L) return 0;
    } else if (c == '+' || c == '-') {
        if (pos + 6u != L) return 0;
        // bytes 1..5 = H H : M M
        uint32_t y0 = (z0 >> 8) ^ 0x003A3030u;   // H H :   (3 bytes)
        uint32_t y1 = (z1 & 0xFFFFu) ^ 0x3030u;  // M M
        if (swar_exceeds(y0, 0x7F7F7676u) | swar_exceeds(y1, 0x7F7F7676u)) return 0;
        p.off_sign = c == '-' ? -1 : 1;
        p.off_h = (int)((y0 & 0xFFu) * 10u + ((y0 >> 8) & 0xFFu));
        p.off_m = (int)((y1 & 0xFFu) * 10u + ((y1 >> 8) & 0xFFu));
    } else {
        return 0;
    }
    return datetime_to_unix(p, true, out) ? 1 : 0;
}

// Fast path of decode() for a line that lives in the LDS tile.  Returns false when the line's
// shape is outside what the fast path recognises (caller then runs parse_line_generic).
__device__ __forceinline__ bool parse_line_fast(const Tile& T, uint32_t base, uint32_t len, Row& r,
                                                const DevTables& t) {
    if (len < 5u) return false;
    // ---- "<" 1-3 digits ">" "1" then ' ' or end of line ------------------------- :62-92
    uint32_t h0, h1;
    load8(T, base, &h0, &h1);
    if ((h0 & 0xFFu) != '<') return false;
    uint32_t c1 = ((h0 >> 8) & 0xFFu) - '0', c2 = ((h0 >> 16) & 0xFFu) - '0', c3 = (h0 >> 24) - '0';
    if (c1 > 9u) return false;
    uint32_t nd = 1, pri = c1;
    if (c2 <= 9u) {
        nd = 2;
        pri = pri * 10u + c2;
        if (c3 <= 9u) {
            nd = 3;
            pri = pri * 10u + c3;
        }
    }
    if (pri > 255u) return false;
    // bytes at 1+nd, 2+nd, 3+nd must be '>', '1', (' ' | end)
    uint64_t hh = (((uint64_t)h1 << 32) | h0) >> (8u * (1u + nd));
    if ((uint32_t)(hh & 0xFFFFu) != (('1' << 8) | '>')) return false;
    const uint32_t sp0 = 3u + nd;
    if (sp0 > len) return false;
    r.facility = pri >> 3;
    r.severity = pri & 7u;
    if (sp0 == len) {
        r.status = E_NOTS;
        return true;
    }
    if ((uint32_t)((hh >> 16) & 0xFFu) != ' ') return false;  // "<13>1x": generic path reports it

    // ---- splitn(7, ' '): positions of the first six spaces from the bitmap ----------- :23
    uint32_t sp[6];
    uint32_t nsp;
    {
        const uint32_t q = base >> 5, sh = base & 31u;
        uint32_t b0 = T.bm[q], b1 = T.bm[q + 1], b2 = T.bm[q + 2], b3 = T.bm[q + 3], b4 = T.bm[q + 4];
        uint64_t lo = (uint64_t)__builtin_amdgcn_alignbit(b1, b0, sh) | ((uint64_t)__builtin_amdgcn_alignbit(b2, b1, sh) << 32);
        uint64_t hi = (uint64_t)__builtin_amdgcn_alignbit(b3, b2, sh) | ((uint64_t)__builtin_amdgcn_alignbit(b4, b3, sh) << 32);
        if (len < 64u) {
            lo &= (1ull << len) - 1ull;
            hi = 0;
        } else if (len < 128u) {
            hi &= (1ull << (len - 64u)) - 1ull;
        }
        nsp = 0;
        uint32_t from = 128u;  // where a continuation beyond the window would resume
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            uint32_t pos = len;
            if (lo) {
                pos = (uint32_t)__builtin_ctzll(lo);
                lo &= lo - 1ull;
                nsp = k + 1;
            } else if (hi) {
                pos = 64u + (uint32_t)__builtin_ctzll(hi);
                hi &= hi - 1ull;
                nsp = k + 1;
            }
            sp[k] = pos;
        }
        if (nsp < 6u && len > 128u) {
            // header longer than the window: continue on the bitmap in LDS (static indices only,
            // so that sp[] stays in registers)
            bool more = true;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                if (more && (uint32_t)k >= nsp) {
                    uint32_t pos = find_space_bm(T, base, from, len);
                    if (pos >= len) {
                        more = false;
                    } else {
                        sp[k] = pos;
                        nsp = k + 1;
                        from = pos + 1u;
                    }
                }
            }
        }
    }
    // sp[0] is the space right after "<PRI>1" by construction (no space can precede it)

    // ---- timestamp ------------------------------------------------------------------- :25
    {
        const uint32_t t0 = sp0 + 1u;
        const uint32_t te = nsp >= 2u ? sp[1] : len;
        int ok = fast_rfc3339(T, base, t0, te - t0, &r.ts);
        if (ok == 2) {
            LdsReader rd(T.w, base);
            ok = parse_rfc3339(rd, t0, te, &r.ts) ? 1 : 0;
        }
        if (!ok) {
            r.status = E_BADTS;
            return true;
        }
    }
    // ---- hostname / appname / procid / msgid ------------------------------------- :26-30
    if (nsp < 6u) {
        r.status = E_NOHOST + (nsp - 1u);  // 1 space -> hostname missing ... 5 -> message data missing
        return true;
    }
    // field k lies between spaces k+1 and k+2 (hostname between the 2nd and 3rd space)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        r.off[k] = sp[k + 1] + 1u;
        r.len[k] = sp[k + 2] - sp[k + 1] - 1u;
    }
    const uint32_t d0 = sp[5] + 1u;
    if (d0 >= len) {
        r.status = E_NOMSG;
        return true;
    }
    // ---- part 7 ------------------------------------------------------------------ :127-161
    uint32_t m0, m1;
    load8(T, base + d0, &m0, &m1);
    const uint32_t c = m0 & 0xFFu;
    if (c != '-') {
        // structured data or garbage: generic tail (byte-walking FSM)
        LdsReader rd(T.w, base);
        parse_tail(rd, d0, 0u, len, r, t);
        return true;
    }
    r.data0 = d0;
    // trim_end of the whole line: common case = last byte is a non-whitespace ASCII char
    uint32_t e = len;
    {
        uint32_t l0, l1;
        load8(T, base + len - 1u, &l0, &l1);
        uint32_t last = l0 & 0xFFu;
        if (!(last > 0x20u && last < 0x80u)) {
            LdsReader rd(T.w, base);
            e = trim_end(rd, 0u, len);
        }
    }
    r.off[S_FULL] = 0;
    r.len[S_FULL] = e;
    // trim_start after the '-': common case = one ' ' then a non-whitespace ASCII char
    uint32_t s = d0 + 1u;
    {
        uint32_t cA = (m0 >> 8) & 0xFFu, cB = (m0 >> 16) & 0xFFu;
        if (s + 1u < len && cA == ' ' && cB > 0x20u && cB < 0x80u) {
            s += 1u;
        } else if (s < len && cA > 0x20u && cA < 0x80u) {
            // "-x": message starts right after the dash
        } else {
            LdsReader rd(T.w, base);
            s = trim_start(rd, s, len);
        }
    }
    if (e > s) {
        r.off[S_MSG] = s;
        r.len[S_MSG] = e - s;
    }
    return true;
}

// One wave per 64-line group.  Dynamic LDS: [tile_cap + 64 bytes of data][bitmap: 2 B per 16 B].
template <int BATCH>
__global__ __launch_bounds__(kWave) void k_rfc5424(const uint8_t* __restrict__ bytes,
                                                  const uint64_t* __restrict__ offsets, uint64_t n,
                                                  DevTables t, uint32_t tile_cap) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint16_t* bm16 = reinterpret_cast<uint16_t*>(smem + tile_cap + 64u);
    const uint32_t lane = threadIdx.x;
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

    // ---- stage A: stream [a0, a0+span) through registers into LDS, classify on the way ----
    {
        const uint4* __restrict__ src = reinterpret_cast<const uint4*>(bytes + a0);
        uint4* dst = reinterpret_cast<uint4*>(smem);
        const uint32_t nchunk = span >> 4;
        for (uint32_t c0 = 0; c0 < nchunk; c0 += kWave * BATCH) {
            uint4 v[BATCH];
#pragma unroll
            for (int k = 0; k < BATCH; ++k) {
                uint32_t idx = c0 + k * kWave + lane;
                if (idx < nchunk) v[k] = src[idx];
            }
#pragma unroll
            for (int k = 0; k < BATCH; ++k) {
                uint32_t idx = c0 + k * kWave + lane;
                if (idx < nchunk) {
                    dst[idx] = v[k];
                    bm16[idx] = (uint16_t)space_mask16(v[k]);
                }
            }
        }
    }
    __syncthreads();

    // ---- stage B: lane-per-line tokenisation ------------------------------------------------
    Row r;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        r.off[k] = 0;
        r.len[k] = FG_NONE;
    }
    const uint32_t len = (uint32_t)(o1 - o0);
    const bool in_tile = (o1 - a0) <= (uint64_t)span;
    const uint32_t base = (uint32_t)(o0 - a0);
    Tile T{reinterpret_cast<const uint32_t*>(smem), reinterpret_cast<const uint32_t*>(bm16)};
    if (valid) {
        bool done = false;
        if (in_tile) done = parse_line_fast(T, base, len, r, t);
        if (!done) {
            // reset whatever the fast path touched before giving up
            r = Row();
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                r.off[k] = 0;
                r.len[k] = FG_NONE;
            }
            if (in_tile) {
                LdsReader rd(T.w, base);
                parse_line_generic(rd, len, r, t);
            } else {
                GlobalReader rd(reinterpret_cast<const uint32_t*>(bytes), o0);
                parse_line_generic(rd, len, r, t);
            }
        }
    }

    // ---- structured-data entries: wave prefix sum + one atomic per wave ---------------------
    uint32_t total;
    uint32_t ex = wave_exclusive_sum(r.n_ent, &total);
    uint32_t first = 0;
    if (total != 0) {  // wave-uniform
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
                uint32_t msg_at, cnt;
                if (in_tile) {
                    LdsReader rd(T.w, base);
                    sd_walk<true>(rd, r.data0, len, &msg_at, &cnt, t, first);
                } else {
                    GlobalReader rd(reinterpret_cast<const uint32_t*>(bytes), o0);
                    sd_walk<true>(rd, r.data0, len, &msg_at, &cnt, t, first);
                }
            }
        }
    }

    // ---- table row: one coalesced store per column ------------------------------------------
    if (valid) {
        const bool ok = r.status == E_OK;
        t.meta[li] = r.status | (r.facility << 8) | (r.severity << 16) | (r.flags << 24);
        t.ts[li] = ok ? r.ts : 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) t.span[k][li] = ok ? fg_span{r.off[k], r.len[k]} : fg_span{0, FG_NONE};
        t.ent_first[li] = first;
        t.ent_count[li] = r.n_ent;
    }
}

}  // namespace fg

// host-side launcher (called from fg_capi.cpp).  tile_cap: multiple of 1024.
extern "C" int fg_launch_rfc5424(const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t n,
                                 const fg::DevTables* t, uint32_t tile_cap, hipStream_t stream) {
    if (n == 0) return 0;
    uint64_t groups = (n + fg::kWave - 1) / fg::kWave;
    if (groups > 0x7FFFFFFFull) return -1;
    uint32_t lds = tile_cap + 64u + (tile_cap / 16u + 16u) * 2u;
    // loads in flight per lane during stage A (tuning knob; 16 B each)
    static int batch = [] {
        const char* e = getenv("FG_BATCH");
        return e ? atoi(e) : 8;
    }();
    dim3 grid((uint32_t)groups), block(fg::kWave);
    switch (batch) {
        case 4: hipLaunchKernelGGL(fg::k_rfc5424<4>, grid, block, lds, stream, d_bytes, d_offsets, n, *t, tile_cap); break;
        case 12: hipLaunchKernelGGL(fg::k_rfc5424<12>, grid, block, lds, stream, d_bytes, d_offsets, n, *t, tile_cap); break;
        case 16: hipLaunchKernelGGL(fg::k_rfc5424<16>, grid, block, lds, stream, d_bytes, d_offsets, n, *t, tile_cap); break;
        case 20: hipLaunchKernelGGL(fg::k_rfc5424<20>, grid, block, lds, stream, d_bytes, d_offsets, n, *t, tile_cap); break;
        default: hipLaunchKernelGGL(fg::k_rfc5424<8>, grid, block, lds, stream, d_bytes, d_offsets, n, *t, tile_cap); break;
    }
    return (int)hipGetLastError();
}
