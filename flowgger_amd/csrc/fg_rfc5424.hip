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
#include <cstdio>
#include <cstdlib>

#include "fg_fused.hpp"
#include "fg_sd2.hpp"
#include "fg_tsfast.hpp"

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
// Stage B: register-resident fast path
// =============================================================================================

// What the straight-line fast path could not finish; each bit sends the lane through one rare,
// branchy piece of the generic parser afterwards.
enum : uint32_t {
    R_GENERIC = 1,  // not "<ddd>1 " shaped, header beyond the 128-byte window: whole line -> parse_line_generic
    R_TS_SLOW = 2,  // leap second, >= 16 fraction digits, result before 1970 / beyond 2^34 s: byte-wise RFC3339
    R_TAIL = 4,     // part 7 does not start with '-': structured data or garbage -> parse_tail
    R_TRIM = 8      // whitespace (possibly multi-byte) at the trim points -> trim_end / trim_start
};

struct Fast {
    uint32_t route;
    uint32_t no_ts;    // "<PRI>1" is the whole line                                    :25
    uint32_t ts_ok;    // timestamp converted on the fast path
    uint32_t rest;     // status of everything after the timestamp (E_OK / E_NOHOST.. / E_NOMSG)
    uint32_t t0, te;   // timestamp part [t0, te)
    uint32_t d0;       // index of part 7
    uint32_t c7;       // its first byte
    uint32_t e, s;     // message end / start after the cheap trims
};

// RFC3339 from registers, branch-free: [t0, t0+L) is the timestamp part.  Returns 1 = converted,
// 0 = invalid, 2 = undecided (the caller runs the byte-wise parser).
// `hdr` = the line's first 132 bytes as line-aligned dwords (already in registers): the timestamp
// starts at t0 in 5..7, i.e. always inside hdr[1].
__device__ __forceinline__ uint32_t fast_rfc3339(const Tile& T, uint32_t base, const uint32_t* hdr, uint32_t t0, uint32_t L,
                                                 double* out) {
    const uint32_t a = base + t0;
    const uint32_t s = t0 - 4u;  // 1..3
    uint32_t r[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) r[k] = __builtin_amdgcn_alignbyte(hdr[k + 2], hdr[k + 1], s);
    return fast_rfc3339_core(r, L, [&](uint32_t pos, uint32_t* z0, uint32_t* z1) { load8(T, a + pos, z0, z1); }, out);
}

// Lowest set bit of the 128-bit window (lo, hi) -> its index, cleared from the window; `none`
// when the window is empty (then *found stays as it is).
__device__ __forceinline__ uint32_t take_space(uint64_t& lo, uint64_t& hi, uint32_t none, uint32_t& found) {
    const bool in_lo = lo != 0, in_hi = hi != 0;
    const uint32_t p = in_lo ? (uint32_t)__builtin_ctzll(lo) : in_hi ? 64u + (uint32_t)__builtin_ctzll(hi) : none;
    const uint64_t lo2 = lo & (lo - 1ull), hi2 = hi & (hi - 1ull);
    hi = in_lo ? hi : hi2;
    lo = lo2;  // (0 & anything == 0: no select needed)
    found += (in_lo || in_hi) ? 1u : 0u;
    return p;
}

// Fast path of decode() for a line that lives in the LDS tile: ONE straight-line block for the
// common shape `<PRI>1 TS HOST APP PROC MSGID - MSG` (every independent LDS read issued up front,
// everything else selects), so that lanes with valid, invalid-but-simple and oddly shaped lines
// stay converged; what it cannot decide is reported in Fast::route.
// (last_byte: the line's last byte when the caller has it -- HEAD staging, where the end of a long line is not in the tile --
//  else kNoLastByte: read from the tile)
constexpr uint32_t kNoLastByte = 0xFFFFFFFFu;
__device__ __forceinline__ Fast parse_line_fast(const Tile& T, uint32_t base, uint32_t len, Row& r, uint32_t last_byte = kNoLastByte) {
    Fast f;
    // ---- independent LDS reads: the line's first 132 bytes (line-aligned dwords) and its last byte.
    //      The space mask of the header is derived HERE, per line, from those registers: classifying
    //      only the ~128 header bytes of each line costs fewer wave-instructions than classifying
    //      every byte of the tile in stage A (most bytes of a log line are message text).
    uint32_t l0 = last_byte, l1;
    if (last_byte == kNoLastByte) load8(T, base + (len ? len - 1u : 0u), &l0, &l1);
    uint32_t hdr[33];
    {
        const uint32_t d = base >> 2, sft = base & 3u;
        uint32_t raw[34];
#pragma unroll
        for (int k = 0; k < 34; ++k) raw[k] = T.w[d + k];
#pragma unroll
        for (int k = 0; k < 33; ++k) hdr[k] = __builtin_amdgcn_alignbyte(raw[k + 1], raw[k], sft);
    }
    const uint32_t h0 = hdr[0], h1 = hdr[1];

    // ---- "<" 1-3 digits ">" "1" then ' ' or end of line ------------------------- :62-92
    const uint32_t c1 = ((h0 >> 8) & 0xFFu) - '0', c2 = ((h0 >> 16) & 0xFFu) - '0', c3 = (h0 >> 24) - '0';
    const bool d2 = c2 <= 9u, d3 = d2 && c3 <= 9u;
    const uint32_t nd = 1u + (d2 ? 1u : 0u) + (d3 ? 1u : 0u);
    uint32_t pri = c1;
    pri = d2 ? pri * 10u + c2 : pri;
    pri = d3 ? pri * 10u + c3 : pri;
    // bytes at 1+nd, 2+nd, 3+nd must be '>', '1', (' ' | end)
    const uint64_t hh = (((uint64_t)h1 << 32) | h0) >> (8u * (1u + nd));
    const uint32_t sp0 = 3u + nd;
    const bool shape = (h0 & 0xFFu) == '<' && c1 <= 9u && pri <= 255u && (uint32_t)(hh & 0xFFFFu) == (('1' << 8) | '>') &&
                       sp0 <= len;
    f.no_ts = sp0 == len ? 1u : 0u;
    // "<13>1x": the generic path reports it
    f.route = (!shape || (sp0 < len && (uint32_t)((hh >> 16) & 0xFFu) != ' ')) ? R_GENERIC : 0u;
    r.facility = pri >> 3;
    r.severity = pri & 7u;

    // ---- splitn(7, ' '): the five spaces after the one that follows "<PRI>1" ----------- :23
    uint64_t lo, hi;
    {
        uint32_t m[8];
#pragma unroll
        for (int k = 0; k < 8; ++k)
            m[k] = gather16(eq_flags(hdr[4 * k], 0x20202020u), eq_flags(hdr[4 * k + 1], 0x20202020u),
                            eq_flags(hdr[4 * k + 2], 0x20202020u), eq_flags(hdr[4 * k + 3], 0x20202020u));
        lo = (uint64_t)(m[0] | (m[1] << 16)) | ((uint64_t)(m[2] | (m[3] << 16)) << 32);
        hi = (uint64_t)(m[4] | (m[5] << 16)) | ((uint64_t)(m[6] | (m[7] << 16)) << 32);
    }
    {
        const uint64_t lo_keep = len < 64u ? (1ull << (len & 63u)) - 1ull : ~0ull;
        const uint64_t hi_keep = len < 64u ? 0ull : len < 128u ? (1ull << ((len - 64u) & 63u)) - 1ull : ~0ull;
        lo &= lo_keep & ~((2ull << sp0) - 1ull);  // no space can precede sp0 in "<ddd>1"; sp0 itself is known
        hi &= hi_keep;
    }
    uint32_t found = 0;
    const uint32_t s1 = take_space(lo, hi, len, found);
    const uint32_t s2 = take_space(lo, hi, len, found);
    const uint32_t s3 = take_space(lo, hi, len, found);
    const uint32_t s4 = take_space(lo, hi, len, found);
    const uint32_t s5 = take_space(lo, hi, len, found);
    // header longer than the window: let the generic parser walk it
    f.route |= (found < 5u && len > 128u) ? R_GENERIC : 0u;

    // ---- timestamp ------------------------------------------------------------------- :25
    f.t0 = sp0 + 1u;
    f.te = s1;
    {
        const uint32_t ok = fast_rfc3339(T, base, hdr, f.t0, s1 - f.t0, &r.ts);
        f.ts_ok = ok == 1u ? 1u : 0u;
        f.route |= ok == 2u ? R_TS_SLOW : 0u;
    }
    // ---- hostname / appname / procid / msgid (verbatim) ------------------------- :26-30
    r.off[S_HOST] = s1 + 1u;
    r.len[S_HOST] = s2 - s1 - 1u;
    r.off[S_APP] = s2 + 1u;
    r.len[S_APP] = s3 - s2 - 1u;
    r.off[S_PROC] = s3 + 1u;
    r.len[S_PROC] = s4 - s3 - 1u;
    r.off[S_MSGID] = s4 + 1u;
    r.len[S_MSGID] = s5 - s4 - 1u;
    const uint32_t d0 = s5 + 1u;
    f.d0 = d0;
    // found = spaces after the first: 0 -> hostname missing ... 4 -> message data missing
    f.rest = found < 5u ? E_NOHOST + found : d0 >= len ? E_NOMSG : E_OK;

    // ---- part 7 ------------------------------------------------------------------ :127-161
    uint32_t m0, m1;
    load8(T, base + (d0 < len ? d0 : 0u), &m0, &m1);
    const uint32_t c = m0 & 0xFFu;
    f.c7 = c;
    const bool live = !f.no_ts && f.rest == E_OK;  // (the timestamp verdict may still come later)
    f.route |= (live && c != '-') ? R_TAIL : 0u;
    // trim_end of the whole line: common case = last byte is a non-whitespace ASCII char
    const uint32_t last = l0 & 0xFFu;
    const bool end_plain = last > 0x20u && last < 0x80u;
    f.e = len;
    // trim_start after the '-': common case = one ' ' then a non-whitespace ASCII char, or "-x"
    const uint32_t cA = (m0 >> 8) & 0xFFu, cB = (m0 >> 16) & 0xFFu;
    const uint32_t s = d0 + 1u;
    const bool one_sp = s + 1u < len && cA == ' ' && cB > 0x20u && cB < 0x80u;
    const bool no_sp = s < len && cA > 0x20u && cA < 0x80u;
    f.s = one_sp ? s + 1u : s;
    f.route |= (live && c == '-' && !(end_plain && (one_sp || no_sp))) ? R_TRIM : 0u;
    return f;
}

// ---------------------------------------------------------------------------------------------
// Structured data out of the tile (the hot form; the byte-walking sd_walk above stays for lines
// outside the tile and for the generic route).
//
// Most SD bytes are VALUE bytes, and inside a value only '"' and '\' matter
// (rfc5424_decoder.rs:206-219).  When a group contains SD lines the wave turns the space bitmap
// -- the fast path is finished with it by then -- into a bitmap of those two characters, and the
// per-lane walker jumps over every value with one bit scan instead of walking it byte by byte;
// names, '=' and separators (a handful of bytes per pair) are still walked.
// ---------------------------------------------------------------------------------------------
struct QuoteClass {  // '"' or '\\'
    static __device__ __forceinline__ uint32_t mask16(const uint4& v) {
        const uint32_t Q = 0x22222222u, B = 0x5C5C5C5Cu;
        return gather16(eq_flags(v.x, Q) | eq_flags(v.x, B), eq_flags(v.y, Q) | eq_flags(v.y, B),
                        eq_flags(v.z, Q) | eq_flags(v.z, B), eq_flags(v.w, Q) | eq_flags(v.w, B));
    }
};
__device__ __forceinline__ bool is_sd_name_char(uint32_t c) {  // :188-192
    return (c - 33u) <= 93u && c != '"' && c != '=' && c != ']';
}
// 16-bit class masks of a 16-byte window (bit i <=> byte i), from exact SWAR byte tests gathered
// with v_dot4_u32_u8.
struct WinMasks {
    uint32_t space; // ' '
    uint32_t skip;  // ' ' or '"'   (the OUT state's ignorable characters, :194,:232)
    uint32_t name;  // 33..=126 minus '"' '=' ']'                                   :188-192
    uint32_t eq, quote, rb;
};
__device__ __forceinline__ WinMasks window_masks(const uint32_t w[4]) {
    uint32_t sp[4], qu[4], eq[4], rb[4], rg[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        sp[k] = eq_flags(w[k], 0x20202020u);
        qu[k] = eq_flags(w[k], 0x22222222u);
        eq[k] = eq_flags(w[k], 0x3D3D3D3Du);
        rb[k] = eq_flags(w[k], 0x5D5D5D5Du);
        rg[k] = range_flags(w[k]);
    }
    WinMasks m;
    m.quote = gather16(qu[0], qu[1], qu[2], qu[3]);
    m.eq = gather16(eq[0], eq[1], eq[2], eq[3]);
    m.rb = gather16(rb[0], rb[1], rb[2], rb[3]);
    m.space = gather16(sp[0], sp[1], sp[2], sp[3]);
    m.skip = m.space | m.quote;
    m.name = gather16(rg[0], rg[1], rg[2], rg[3]) & ~(m.quote | m.eq | m.rb);
    return m;
}

// Same contract as sd_walk, for a line in the tile whose group has the quote/backslash bitmap
// (T.bm) built.  Lane-per-line, but token- instead of byte-granular: a serial byte walk on a GPU
// costs one dependent LDS round trip (hundreds of cycles) per byte; here ONE pair costs two:
//   round 1  bit scan of the quote bitmap from the value's first byte -> p (closing '"' or a '\');
//   round 2  a 16-byte window AT p: byte 0 says which of the two it is, bytes 1.. hold the next
//            pair's "<spaces>name=\"" prefix, resolved in registers from SWAR class masks.
//
// MODE  SD_COUNT  count entries only
//       SD_EMIT   write them to the entry table starting at `slot`
//       SD_STASH  count AND park a compact 64-bit record per entry in the wave's scratch
//                 (stash[k * 64 + lane], k < kStashEntries): once the wave has its slots, the
//                 entries are copied out of the stash instead of parsing every line a second time.
enum { SD_COUNT = 0, SD_EMIT = 1, SD_STASH = 2 };
// record: name_s | name_len << 16 | val_len << 32 | esc << 48 | is_sdid << 49   (val_s = name_s + name_len + 2)
__device__ __forceinline__ uint64_t stash_pack(uint32_t name_s, uint32_t name_len, uint32_t val_len, uint32_t esc, uint32_t sdid) {
    return (uint64_t)name_s | ((uint64_t)name_len << 16) | ((uint64_t)val_len << 32) | ((uint64_t)esc << 48) | ((uint64_t)sdid << 49);
}
// SD_STASH: the records go INTO THE LINE ITSELF -- the tile bytes the walk has already consumed (header, earlier pairs) are dead, names
// and values are recorded as offsets -- as 8-byte records from the line's first 4-byte boundary on (round 2 parked them in a stash
// in global memory: 224 B per line of HBM traffic on the 13-pair corpus).  A record that would reach into bytes still to be read
// (many tiny pairs: `a="" b="" ...`) clears *rec_ok: the caller then writes the line's entries by a second walk over the line in
// GLOBAL memory (its copy in the tile is no longer intact).
template <int MODE>
__device__ __forceinline__ uint32_t sd_walk_tile(const Tile& T, uint32_t base, uint32_t pos, uint32_t len, uint32_t* msg_at,
                                                 uint32_t* n_ent, const DevTables& t, uint32_t slot, uint32_t* tile_w = nullptr,
                                                 bool* rec_ok_out = nullptr) {
    constexpr bool EMIT = MODE == SD_EMIT;
    LdsReader rd(T.w, base);
    uint32_t cnt = 0;
    uint32_t wpos = (base + 3u) & ~3u;  // tile byte of the next record
    bool rec_ok = true;
    auto record = [&](uint64_t rec, uint32_t consumed) {  // consumed: line index up to which the walk is done with the bytes
        if (rec_ok) {
            if (wpos + 8u <= base + consumed) {
                tile_w[wpos >> 2] = (uint32_t)rec;
                tile_w[(wpos >> 2) + 1u] = (uint32_t)(rec >> 32);
                wpos += 8u;
            } else {
                rec_ok = false;
            }
        }
    };
    for (;;) {
        // sd_id = bytes after '[' up to the first ' ' (anything allowed)            :175-177
        const uint32_t s = pos + 1;
        uint32_t sp = s;
        for (;;) {  // 16 bytes per step
            if (sp >= len) return E_NOSD;
            uint32_t w[4];
            load16(T, base + sp, w);
            const uint32_t avail = len - sp < 16u ? len - sp : 16u;
            const uint32_t hit = gather16(eq_flags(w[0], 0x20202020u), eq_flags(w[1], 0x20202020u), eq_flags(w[2], 0x20202020u),
                                          eq_flags(w[3], 0x20202020u)) & ((1u << avail) - 1u);
            if (hit) {
                sp += (uint32_t)__builtin_ctz(hit);
                break;
            }
            sp += avail;
        }
        if (EMIT) {
            t.ent_name[slot + cnt] = fg_span{s, sp - s};
            t.ent_val[slot + cnt] = 0;
            t.ent_type[slot + cnt] = FG_T_SDID;
            t.ent_flags[slot + cnt] = 0;
        }
        if (MODE == SD_STASH) record(stash_pack(s, sp - s, 0, 0, 1), sp + 1u);
        ++cnt;
        uint32_t status = E_OK;
        uint32_t i = sp + 1;       // OUT state: next unread byte
        bool in_value = false;     // true: [val_s, ...) is an open value, `cur` = where to look for its end
        uint32_t name_s = 0, name_e = 0, val_s = 0, cur = 0, esc_seen = 0;
        uint32_t close_at = 0;     // index of the element's ']'
        for (;;) {
            uint32_t w0 = i, start = 0;
            if (in_value) {
                const uint32_t p = find_bit(T.bm, base, cur, len);
                if (p >= len) {
                    status = E_NOBRACKET;  // input exhausted inside a value                  :239
                    break;
                }
                w0 = p;
                start = 1;
            } else if (i >= len) {
                status = E_NOBRACKET;
                break;
            }
            uint32_t w[4];
            load16(T, base + w0, w);
            if (in_value) {
                if ((w[0] & 0xFFu) == '\\') {  // escapes the next char, whatever it is   :207-213
                    esc_seen = 1;
                    cur = w0 + 2;
                    continue;
                }
                // closing quote: the pair is complete                                      :214-228
                if (EMIT) {
                    t.ent_name[slot + cnt] = fg_span{name_s, name_e - name_s};
                    t.ent_val[slot + cnt] = (uint64_t)val_s | ((uint64_t)(w0 - val_s) << 32);
                    t.ent_type[slot + cnt] = FG_T_STRING;
                    t.ent_flags[slot + cnt] = esc_seen ? FG_EF_VAL_ESC : 0;
                }
                if (MODE == SD_STASH) record(stash_pack(name_s, name_e - name_s, w0 - val_s, esc_seen, 0), w0 + 1u);
                ++cnt;
                in_value = false;
            }
            // ---- OUT state at window offset `start`: <skip chars> then ']' | name '=' '"' -----
            const uint32_t avail = len - w0 < 16u ? len - w0 : 16u;  // window bytes inside the line (>= 1)
            const uint32_t inside = (1u << avail) - 1u;                // avail <= 16
            const WinMasks m = window_masks(w);
            const uint32_t from = ~((1u << start) - 1u);
            const uint32_t stop = ~m.skip & from & inside;             // first byte that is not ' ' / '"'
            if (stop == 0) {                                           // only ignorable bytes in view
                i = w0 + avail;
                continue;                                              // (i >= len is caught at the top)
            }
            const uint32_t i0 = (uint32_t)__builtin_ctz(stop);
            if ((m.rb >> i0) & 1u) {
                close_at = w0 + i0;                                    // unescaped ']' outside name/value :197
                break;
            }
            if (!((m.name >> i0) & 1u)) {
                status = E_SDFMT;                                      //                                   :235
                break;
            }
            // name = run of name chars from i0; then '=' and '"' must follow, all inside the view
            const uint32_t after_name = ~m.name & ~((1u << i0) - 1u) & 0xFFFFu;
            const uint32_t e0 = after_name ? (uint32_t)__builtin_ctz(after_name) : 16u;
            if (e0 + 1u < avail) {
                if (!((m.eq >> e0) & 1u) || !((m.quote >> (e0 + 1u)) & 1u)) {
                    status = E_SDFMT;
                    break;
                }
                name_s = w0 + i0;
                name_e = w0 + e0;
                val_s = name_e + 2u;
            } else if (i0 != 0) {
                i = w0 + i0;                                           // re-window with the name at offset 0
                continue;
            } else {
                // a name of 14+ bytes (or the line ends inside this prefix): byte-wise
                uint32_t q = w0, c = 0;
                do {
                    ++q;
                    c = q < len ? rd.byte(q) : 0x100u;
                } while (is_sd_name_char(c));
                if (q >= len || q + 1u >= len) {
                    // exhausted in IN_NAME / HAVE_NAME (a non-'=' / non-'"' byte there is a format error first)
                    status = (q < len && c != '=') ? E_SDFMT : E_NOBRACKET;
                    break;
                }
                if (c != '=' || rd.byte(q + 1u) != '"') {
                    status = E_SDFMT;
                    break;
                }
                name_s = w0;
                name_e = q;
                val_s = q + 2u;
            }
            in_value = true;
            esc_seen = 0;
            cur = val_s;
        }
        if (status != E_OK) return status;
        const uint32_t after = close_at + 1;
        if (after >= len) return E_NOMSG;  // :148
        const uint32_t c = rd.byte(after);
        if (c == '[') {
            pos = after;
            continue;
        }
        if (c != ' ') return E_MALFORMED;  // :154
        *msg_at = after;
        *n_ent = cnt;
        if (rec_ok_out) *rec_ok_out = rec_ok;
        return E_OK;
    }
}
// parse_tail for a line in the tile whose part 7 starts with '[' and whose group has the
// quote bitmap built.
// walk_len < len (HEAD staging): only the line's first walk_len bytes are in the tile.  The walk runs over them as if the line
// ended there; when it closes the structured data inside them, and the trims are the everyday ones (one ASCII non-blank byte at
// each end: the line's last byte comes in last2), the result does not depend on the bytes that are not there.  Anything else
// returns false: the caller parses the line from global memory.
// the trims behind a structured-data walk that ended at msg_at (parse_msg :163-172, full_msg :46); false = the head of the line was
// not enough (HEAD staging): the caller parses the line from global memory
__device__ __forceinline__ bool finish_sd_tail(const Tile& T, uint32_t base, uint32_t len, uint32_t walk_len, uint32_t last2, uint32_t msg_at, Row& r,
                                               const uint8_t* gbytes, uint64_t o0) {
    LdsReader rd(T.w, base);
    if (walk_len < len) {
        // the trims: the message starts in the head (a run of whitespace that reaches its end is not decided here); the line's end is
        // its last byte when that is a plain one, else the few bytes behind it are looked at where they are, in global memory
        const uint32_t s = trim_start(rd, msg_at, walk_len - 8u);
        if (s + 8u >= walk_len) return false;
        const uint32_t last = last2 & 0xFFu;
        uint32_t e = len;
        if (!(last > 0x20u && last < 0x80u)) {
            GlobalReader grd(reinterpret_cast<const uint32_t*>(gbytes), o0);
            e = trim_end(grd, 0u, len);
        }
        r.off[S_FULL] = 0;
        r.len[S_FULL] = e;
        if (e > s) {
            r.off[S_MSG] = s;
            r.len[S_MSG] = e - s;
        }
        return true;
    }
    uint32_t e = trim_end(rd, 0u, len);
    r.off[S_FULL] = 0;
    r.len[S_FULL] = e;
    uint32_t s = trim_start(rd, msg_at, len);
    if (e > s) {
        r.off[S_MSG] = s;
        r.len[S_MSG] = e - s;
    }
    return true;
}
__device__ __forceinline__ bool parse_tail_sd_tile(const Tile& T, uint32_t base, uint32_t q, uint32_t len, uint32_t walk_len, uint32_t last2, Row& r,
                                                   const DevTables& t, uint32_t* tile_w, bool* rec_ok, const uint8_t* gbytes, uint64_t o0) {
    r.data0 = q;
    uint32_t msg_at = 0;
    uint32_t st = tile_w ? sd_walk_tile<SD_STASH>(T, base, q, walk_len, &msg_at, &r.n_ent, t, 0, tile_w, rec_ok)
                         : sd_walk_tile<SD_COUNT>(T, base, q, walk_len, &msg_at, &r.n_ent, t, 0);
    if (st != E_OK) {
        r.n_ent = 0;
        if (walk_len < len) return false;
        r.status = st;
        return true;
    }
    if (!finish_sd_tail(T, base, len, walk_len, last2, msg_at, r, gbytes, o0)) {
        r.n_ent = 0;
        return false;
    }
    return true;
}

constexpr uint32_t kShortSdTail = 64;  // bytes after the header up to which a lone SD tail is walked byte-wise

// The format policy of the streaming pipeline (fg_pipeline.hpp): stage A builds the SPACE bitmap;
// decode() = stage B + SD entries + the table row for ONE line group whose tile is in LDS.
// SDX = true: the instantiation for batches of lines long enough to carry structured data (the launcher: average >= 320 bytes): the LDS
// behind the tile holds a second bitmap and the scratch of the pair-parallel walk (fg_sd2.hpp); structured data takes that walk, and
// the (rare) lines it hands back are parsed from global memory.  SDX = false is byte for byte the kernel of round 3.
template <bool HEAD, bool SDX = false, bool PROF = false>
struct Rfc5424FormatT {
    // no stage-A bitmap: the fast path classifies the header bytes itself, the SD walker's
    // quote bitmap is built on demand (rebuild_bitmap) for groups that hold SD lines.  (Round 5 measured the pair-parallel kernel with
    // the quote / backslash masks computed in stage A, while the bytes sit in registers, so that the chunk pass no longer reads the tile
    // back: 1.88-1.91 G lines/s against 1.90-1.92 G without, alternated on one box -- profiles/r05d_ab_stagea_cfg4.log: the chunk pass's
    // LDS read is not what the group waits for.  Not kept.)
    static constexpr uint32_t kClasses = 0;
    static __device__ __forceinline__ void classify_store(const uint4&, uint16_t*, uint32_t, uint32_t, uint32_t) {}
    unsigned long long* pacc = nullptr;  // measurement build: the wave's phase clocks (LDS), flushed by the kernel at its end

    __device__ __forceinline__ RowOut decode(const GroupCtx& c, const DevTables& t) const {
    const uint8_t* __restrict__ bytes = c.bytes;
    const uint8_t* smem = c.smem;
    uint16_t* bm16 = c.bm16;
    const uint64_t o0 = c.o0, o1 = c.o1, a0 = c.a0;
    const uint32_t span = c.span, ablate = c.ablate;
    const bool valid = c.valid;
    uint64_t* stash = c.stash;
    const uint32_t lane = threadIdx.x;
    Row r;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        r.off[k] = 0;
        r.len[k] = FG_NONE;
    }
    const uint32_t len = (uint32_t)(o1 - o0);
    // HEAD staging: the tile holds the line's first c.tlen bytes at c.tbase -- the whole line (in_tile) or its head only
    const bool in_tile = HEAD ? c.tlen == len : (o1 - a0) <= (uint64_t)span;
    const bool head_only = HEAD && !in_tile && c.tlen >= 512u;
    const uint32_t walk_len = HEAD ? c.tlen : len;
    const uint32_t base = HEAD ? c.tbase : (uint32_t)(o0 - a0);
    bool from_global = false;  // the line ended up parsed from global memory (its entries are written from there as well)
    uint32_t* tile_w = reinterpret_cast<uint32_t*>(const_cast<uint8_t*>(smem));
    bool rec_ok = false;       // the structured-data walk left the line's entries as records in the line's own tile bytes
    Tile T{reinterpret_cast<const uint32_t*>(smem), reinterpret_cast<const uint32_t*>(bm16)};
    // measurement build: cycles of  0 header fast path  1 chunk pass  2-5 the pair-parallel walk (fg_sd2.hpp)  6 routes + trims
    //                               7 entry slots  8 emit / entry stores  9 row
    uint64_t pc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tk = PROF ? wv::clock() : 0;
    auto tick = [&](int k) {
        if (PROF) {
            const uint64_t now = wv::clock();
            pc[k] += now - tk;
            tk = now;
        }
    };
    // ---- 1. straight-line fast path for every line of the tile --------------------------------
    uint32_t route = 0;
    Fast f{};
    if (valid) {
        route = R_GENERIC;
        if (in_tile || head_only) {
            f = parse_line_fast(T, base, len, r, HEAD ? (len ? (c.last2 & 0xFFu) : 0u) : kNoLastByte);
            route = f.route;
            if (!(route & R_GENERIC)) {
                if (route & R_TS_SLOW) {  // rare
                    LdsReader rd(T.w, base);
                    bool ok = parse_rfc3339(rd, f.t0, f.te, &r.ts);
                    r.status = f.no_ts ? E_NOTS : ok ? f.rest : E_BADTS;
                } else {
                    r.status = f.no_ts ? E_NOTS : f.ts_ok ? f.rest : E_BADTS;
                }
            }
        }
    }
    // ---- 2. structured data in this group?  Then the space bitmap (no longer needed) becomes the
    //         quote/backslash bitmap, built by the whole wave -------------------------------------
    // (a group with only a few SHORT bracketed tails -- typically malformed lines of a corpus without structured
    //  data -- is not worth two barriers and a pass over the whole tile: those lanes take the byte-wise parse_tail)
    const bool sd_any = valid && !(route & R_GENERIC) && (route & R_TAIL) && r.status == E_OK && f.c7 == '[';
    const uint64_t sd_ballot = __ballot(sd_any);
    const bool group_has_sd = __any(sd_any && len - f.d0 > kShortSdTail) || __popcll(sd_ballot) > 4;  // wave-uniform
    const bool sd_lane = sd_any && group_has_sd;
    tick(0);
    sd2::Lds SL{};
    sd2::LineIn sin{false, 0u, 0u, 0u, false};
    sd2::LineOut slo{false, 0u, false, E_OK, 0u, 0u, 0u};
    bool sd2_ran = false;  // wave-uniform
    if constexpr (SDX) {
        if (group_has_sd) {
            const uint32_t stride16 = c.tile_cap / 16u + 16u;
            SL = sd2::carve(smem, bm16, c.tile_cap, reinterpret_cast<uint8_t*>(bm16 + 2u * stride16));
            __syncthreads();
            const bool chain = sd2::classify_tile(SL, span);
            __syncthreads();
            tick(1);
            sin = sd2::LineIn{sd_lane && (in_tile || head_only), base, f.d0, walk_len, in_tile};
            if (!chain) {
                slo = sd2::group_walk<PROF>(SL, span, sin, pc);
                sd2_ran = true;
            }
            tk = PROF ? wv::clock() : 0;
        }
    } else if (group_has_sd) {
        __syncthreads();
        rebuild_bitmap<QuoteClass>(smem, bm16, span >> 4);
        __syncthreads();
    }
    // ---- 3. the rare / heavy routes ------------------------------------------------------------
    bool redo = false;  // (HEAD) the head of the line was not enough: the whole line again, from global memory
    bool coop = false;  // (HEAD) ... by the whole wave, once this group is done (the line fits the tile)
    if (valid) {
        if (!(route & R_GENERIC)) {
            if (r.status == E_OK) {
                if (SDX && sd_lane) {
                    // the pair-parallel walk has the line's verdict, or hands it back: then the whole line again, from global memory
                    // (its copy in the tile carries the walk's bookkeeping in the header bytes)
                    redo = !slo.handled;
                    if (slo.handled) {
                        r.data0 = f.d0;
                        if (slo.status != E_OK) {
                            r.status = slo.status;
                        } else {
                            r.n_ent = slo.n_ent;
                            redo = !finish_sd_tail(T, base, len, walk_len, c.last2, slo.msg_at, r, bytes, o0);
                        }
                    }
                } else if (sd_lane) {
                    redo = !parse_tail_sd_tile(T, base, f.d0, len, walk_len, c.last2, r, t, stash ? tile_w : nullptr, &rec_ok, bytes, o0);
                } else if (head_only && (route & R_TAIL)) {
                    redo = true;  // (a short bracketed tail or garbage in a long line)
                } else if (route & R_TAIL) {  // garbage instead of '-' / '['
                    LdsReader rd(T.w, base);
                    parse_tail(rd, f.d0, 0u, len, r, t);
                } else {
                    uint32_t e = f.e, s0 = f.s;
                    if (route & R_TRIM) {  // rare
                        LdsReader rd(T.w, base);
                        if (head_only) {  // the message starts in the head; its end is looked at in global memory
                            s0 = trim_start(rd, f.d0 + 1u, walk_len - 8u);
                            GlobalReader grd(reinterpret_cast<const uint32_t*>(bytes), o0);
                            e = trim_end(grd, 0u, len);
                            redo = s0 + 8u >= walk_len;
                        } else {
                            e = trim_end(rd, 0u, len);
                            s0 = trim_start(rd, f.d0 + 1u, len);
                        }
                    }
                    r.data0 = f.d0;
                    r.off[S_FULL] = 0;
                    r.len[S_FULL] = e;
                    r.off[S_MSG] = e > s0 ? s0 : 0u;
                    r.len[S_MSG] = e > s0 ? e - s0 : FG_NONE;
                }
            }
        } else {  // rare: anything the fast path does not recognise, or a line outside the tile
            redo = !in_tile;
            if (in_tile) {
                r = Row();
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    r.off[k] = 0;
                    r.len[k] = FG_NONE;
                }
                LdsReader rd(T.w, base);
                parse_line_generic(rd, len, r, t);
            }
        }
        if (HEAD && redo && (o1 - (o0 & ~15ull)) <= (uint64_t)c.tile_cap) {
            // (round 5) the line fits the tile WHOLE: it is decoded again at the end of this group, by the whole wave, out of a
            // restaged tile (coop_redo below) -- not by this one lane byte for byte through global memory
            coop = true;
            redo = false;
            slo.handled = false;
            r = Row();
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                r.off[k] = 0;
                r.len[k] = FG_NONE;
            }
        }
        if (redo) {
            slo.handled = false;  // (its entries are written by the global walk below, not by the pair lanes)
            from_global = true;
            r = Row();
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                r.off[k] = 0;
                r.len[k] = FG_NONE;
            }
            GlobalReader rd(reinterpret_cast<const uint32_t*>(bytes), o0);
            parse_line_generic(rd, len, r, t);
        }
    }

    tick(6);
    // ---- structured-data entries: wave prefix sum + one atomic per wave ---------------------
    uint32_t first = 0;
    {
        const EntAlloc ea = alloc_entries_ex(t, r.n_ent, c.ent_state);
        tick(7);
        const uint32_t mine = (ea.overflow || ea.total == 0u) ? 0u : ea.s.at(ea.ex);
        if (ea.overflow) {
            r.status = FG_ST_OVERFLOW;
            r.n_ent = 0;
        }
        first = r.n_ent != 0 ? mine : 0u;
        const bool by_pairs = SDX && sd_lane && slo.handled && !from_global;  // this lane's entries are written by the pair lanes
        if constexpr (SDX) {
            if (sd2_ran) sd2::group_emit(SL, t, sin, slo, by_pairs && !ea.overflow && !(ablate & 4u), first);
        }
        if (r.n_ent != 0 && !(ablate & 4u) && !by_pairs) {
            uint32_t msg_at, cnt;
            if (sd_lane && !from_global && stash && rec_ok) {
                // the records sit in the line's own (consumed) bytes: four in flight before the first store
                const uint32_t* rec32 = tile_w + (((base + 3u) & ~3u) >> 2);
                for (uint32_t k0 = 0; k0 < r.n_ent; k0 += 4u) {
                    uint32_t lo[4], hi[4];
#pragma unroll
                    for (uint32_t j = 0; j < 4u; ++j) {
                        const uint32_t k = k0 + j < r.n_ent ? k0 + j : r.n_ent - 1u;
                        lo[j] = rec32[k * 2u];
                        hi[j] = rec32[k * 2u + 1u];
                    }
#pragma unroll
                    for (uint32_t j = 0; j < 4u; ++j) {
                        const uint32_t k = k0 + j;
                        if (k < r.n_ent) {
                            const uint32_t name_s = lo[j] & 0xFFFFu, name_len = lo[j] >> 16, val_len = hi[j] & 0xFFFFu;
                            const bool sdid = (hi[j] >> 17) & 1u;
                            t.ent_name[first + k] = fg_span{name_s, name_len};
                            t.ent_val[first + k] = sdid ? 0ull : ((uint64_t)(name_s + name_len + 2u) | ((uint64_t)val_len << 32));
                            t.ent_type[first + k] = sdid ? FG_T_SDID : FG_T_STRING;
                            t.ent_flags[first + k] = ((hi[j] >> 16) & 1u) ? FG_EF_VAL_ESC : 0;
                        }
                    }
                }
            } else if (from_global || (sd_lane && stash)) {
                // parsed from global memory -- or a line whose records did not fit its consumed bytes, whose copy in the tile is
                // therefore no longer intact
                GlobalReader rd(reinterpret_cast<const uint32_t*>(bytes), o0);
                sd_walk<true>(rd, r.data0, len, &msg_at, &cnt, t, first);
            } else if (sd_lane) {  // (measurement build without records: the tile is intact)
                sd_walk_tile<SD_EMIT>(T, base, r.data0, walk_len, &msg_at, &cnt, t, first);
            } else if (in_tile) {
                LdsReader rd(T.w, base);
                sd_walk<true>(rd, r.data0, len, &msg_at, &cnt, t, first);
            } else {
                GlobalReader rd(reinterpret_cast<const uint32_t*>(bytes), o0);
                sd_walk<true>(rd, r.data0, len, &msg_at, &cnt, t, first);
            }
        }
    }

    tick(8);
    // ---- table row (stored by the caller: one coalesced store per column) --------------------
    RowOut o;
    const bool ok = r.status == E_OK;
    o.meta = r.status | (r.facility << 8) | (r.severity << 16) | (r.flags << 24);
    o.ts = ok ? r.ts : 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) o.span[k] = ok ? fg_span{r.off[k], r.len[k]} : fg_span{0, FG_NONE};
    o.first = first;
    o.count = r.n_ent;
    if constexpr (HEAD) {
        // ---- lines whose head was not enough (structured data that runs past it, a bracketed tail behind it ...): one in five hundred
        // of the 64 B .. 8 KiB corpus.  Rounds 3-4 sent the lane through the byte-wise parser over global memory, a dword per trip:
        // ~0.65 ms for a 1 KiB structured-data block while the other 63 lanes waited -- THE floor of a small batch's kernel time (64 K
        // lines: 781 us with those lines, 124 us without, profiles/r05a_small*.log) and a third of a large one's.  Now the wave takes
        // such a line as a group of its own: the WHOLE line is staged (coalesced, as stage A would) and decoded by the instantiation
        // for whole lines -- pair-parallel walk included -- and the row goes back to the lane that owns the line.
        unsigned long long todo = __ballot(coop);
        while (todo) {  // wave-uniform, rare
            const int src = (int)__builtin_ctzll(todo);
            todo &= todo - 1ull;
            auto rl64 = [&](uint64_t x) -> uint64_t {
                return (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, src) |
                       ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(x >> 32), src) << 32);
            };
            const uint64_t ro0 = rl64(o0), ro1 = rl64(o1), rli = rl64(c.li);
            const uint64_t ra0 = ro0 & ~15ull;
            const uint32_t rspan = (uint32_t)((ro1 - ra0 + 15ull) & ~15ull);  // <= tile_cap (a multiple of 1024)
            __syncthreads();  // every lane is done with the tile
            {
                __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(bytes + ra0), (short)0, (int)rspan, 0x00020000);
                uint4* dst = reinterpret_cast<uint4*>(const_cast<uint8_t*>(smem));
                const uint32_t nrow = (rspan + 1023u) >> 10;
                for (uint32_t r0 = 0; r0 < nrow; r0 += 4u) {  // (rows past the span fetch nothing and store zeros, as in stage A)
                    u32x4 w[4];
#pragma unroll
                    for (uint32_t j = 0; j < 4u; ++j) w[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(lane * 16u + (r0 + j) * 1024u), 0, FG_STREAM_AUX);
#pragma unroll
                    for (uint32_t j = 0; j < 4u; ++j)
                        if (r0 + j < nrow) dst[(r0 + j) * kWave + lane] = make_uint4(w[j][0], w[j][1], w[j][2], w[j][3]);
                }
            }
            __syncthreads();
            GroupCtx c2{bytes, smem, bm16, ro0, ro1, ra0, rspan, lane == 0u, rli, c.stash, ablate, c.ent_state, nullptr};
            c2.tile_cap = c.tile_cap;
            Rfc5424FormatT<false, SDX, false> whole;
            const RowOut w = whole.decode(c2, t);
            // lane 0's row -> the lane that owns the line
            auto b32 = [&](uint32_t x) -> uint32_t { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); };
            const uint32_t g_meta = b32(w.meta), g_first = b32(w.first), g_count = b32(w.count);
            const uint64_t tsb = (uint64_t)__builtin_bit_cast(unsigned long long, w.ts);
            const uint32_t g_ts0 = b32((uint32_t)tsb), g_ts1 = b32((uint32_t)(tsb >> 32));
            uint32_t g_off[6], g_len[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                g_off[k] = b32(w.span[k].off);
                g_len[k] = b32(w.span[k].len);
            }
            if ((int)lane == src) {
                o.meta = g_meta;
                o.ts = __builtin_bit_cast(double, (unsigned long long)((uint64_t)g_ts0 | ((uint64_t)g_ts1 << 32)));
#pragma unroll
                for (int k = 0; k < 6; ++k) o.span[k] = fg_span{g_off[k], g_len[k]};
                o.first = g_first;
                o.count = g_count;
            }
        }
    }
    tick(9);
    if (PROF && pacc && lane == 0)
        for (int k = 0; k < 10; ++k) pacc[k] += (unsigned long long)pc[k];  // (the wave's own LDS words: a hot global atomic per phase WOULD BE the profile)
    return o;
    }
};

// PROF = true is a measurement build of the same kernel (s_memtime stamps at the phase boundaries,
// FG_PROF=1 / FG_ABLATE, see fg_pipeline.hpp); never the product path.
using Rfc5424Format = Rfc5424FormatT<false>;

// HEAD = true: the instantiation for LONG lines (only the head of every line is staged, fg_pipeline.hpp)
template <int NB, bool PROF, bool HEAD = false, bool SDX = false>
__global__ __launch_bounds__(kWave, 2) void k_rfc5424(const uint8_t* __restrict__ bytes,
                                                     const uint64_t* __restrict__ offsets, uint64_t n, DevTables t,
                                                     uint32_t tile_cap, uint32_t L, uint64_t groups,
                                                     unsigned long long* prof, uint64_t* stash_base, FrameArgs fr) {
    Rfc5424FormatT<HEAD, SDX, PROF> fmt;
    if constexpr (PROF) {
        __shared__ unsigned long long pacc[10];
        if (threadIdx.x < 10) pacc[threadIdx.x] = 0ull;
        fmt.pacc = pacc;
        __syncthreads();
        persistent_loop<NB, PROF, Rfc5424FormatT<HEAD, SDX, PROF>, HEAD>(bytes, offsets, n, t, tile_cap, L, groups, prof, stash_base, fmt, fr);
        __syncthreads();
        if (prof && threadIdx.x < 10) atomicAdd(&prof[6 + threadIdx.x], pacc[threadIdx.x]);
    } else if constexpr (SDX) {
        // (the pair-parallel kernel keeps the tables' forty words in LDS: in scalar registers they push the kernel past the 102 it has,
        //  and the compiler parks whole kernel-argument tuples in VGPR lanes -- see k_gelf)
        __shared__ DevTables t_lds;
        if (threadIdx.x == 0) t_lds = t;
        __syncthreads();
        persistent_loop<NB, PROF, Rfc5424FormatT<HEAD, SDX, PROF>, HEAD>(bytes, offsets, n, t_lds, tile_cap, L, groups, prof, stash_base, fmt, fr);
    } else {
        persistent_loop<NB, PROF, Rfc5424FormatT<HEAD, SDX, PROF>, HEAD>(bytes, offsets, n, t, tile_cap, L, groups, prof, stash_base, fmt, fr);
    }
}

// The same decoder over a RAW stream: the kernel frames its tiles itself (fg_fused.hpp).  Whole lines only (a stream of long lines
// -- head staging -- keeps the separate framing pass).
template <int NB, bool SDX>
__global__ __launch_bounds__(kWave, 2) void k_rfc5424_fused(const uint8_t* __restrict__ bytes, DevTables t, uint32_t tile_cap, uint32_t L,
                                                           uint64_t* stash_base, FusedArgs fa, uint32_t strip) {
    Rfc5424FormatT<false, SDX, false> fmt;
    // (the tables AND the launch's arguments live in LDS: in scalar registers the loop's own wave-uniform state pushes the kernel past
    //  the 102 it has, whole tuples get parked in VGPR lanes, the VGPRs spill -- and a scratch reload waits for EVERY load in flight)
    __shared__ DevTables t_lds;
    __shared__ FusedArgs fa_lds;
    if (threadIdx.x == 0) {
        t_lds = t;
        fa_lds = fa;
    }
    __syncthreads();
    fused_loop<NB>(bytes, t_lds, tile_cap, L, fmt, fa_lds, strip, stash_base);
}

}  // namespace fg

extern "C" uint64_t fg_stash_bytes(uint32_t blocks) {
    return (uint64_t)blocks * fg::kStashEntries * fg::kStashWords * fg::kWave * sizeof(uint64_t);
}

// host-side launcher (called from fg_capi.cpp).  stash: device scratch of fg_stash_bytes(stash_blocks)
// bytes (or NULL: SD lines are then parsed twice); the persistent grid is capped at stash_blocks.
extern "C" int fg_launch_rfc5424(const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t n, const fg::DevTables* t,
                                 uint64_t avg_len, hipStream_t stream, uint64_t* stash, uint32_t stash_blocks, uint32_t strip,
                                 const uint8_t* line_bad, const fg_launch_opts* lo, fg::TicketSlot* tk) {
    if (n == 0) return 0;
    fg::LaunchPlan p;
    // long lines: only the head of every line is staged (persistent_loop<..., HEAD>): the message is never looked into
    const bool head = (lo->flags & FG_LO_FORCE_HEAD) || (avg_len >= 768u && !(lo->flags & FG_LO_NO_HEAD));
    // lines long enough to carry structured data: the instantiation with the pair-parallel walk (its LDS scratch costs the headline
    // configuration -- 254-byte lines without structured data -- a wave per CU, so short lines keep the kernel they had)
    const bool sdx = (lo->flags & FG_LO_SD_PAIRS) || (avg_len >= 320u && !(lo->flags & FG_LO_SD_WALK));
    const uint64_t plan_len = head ? (avg_len < fg::kHeadCap ? avg_len : fg::kHeadCap) : avg_len;
    const uint32_t sb = stash ? stash_blocks : 0u;
    int prc;
    if (sdx)
        // (default tile from the same-box sweeps, profiles/r04g_sweep_*: 12 KiB -- eight waves per CU -- for whole lines and for heads)
        // (tiles of at most 36 KiB: with the two bitmaps and the walk's scratch -- 0.57 x the tile -- that is 58 KiB of LDS per wave)
        // (THREE waves per SIMD -- the compiler fits the whole-line instantiation into 168 registers without a spill under
        //  __launch_bounds__(64, 3) -- with the 8 KiB tile and window that twelve waves per CU leave each other: 1.64-1.71 G lines/s
        //  against 1.98-2.08 at 4-16 M lines, 9 KiB / eleven waves 1.71-1.78: fourteen lines to the group instead of twenty-one, and a
        //  group's cost is mostly fixed -- profiles/r05x_sd_three_waves_ab.log)
        // (chunks of 128 lines, drawn by ticket from two per wave on: round 4's 1024 was the round-robin form's optimum; under tickets,
        //  one box, alternated -- profiles/r05u_chunk_taper_sweep.log -- 4 M lines: 1879-1905 M lines/s as one share per wave, 1990 in
        //  chunks of 128, 1938-1967 at 256 / 512; 16 M lines: 2008-2013 at 1024, 2074-2082 at 128 .. 512; the long-tail corpus 1349 -> 1396.
        //  From FOUR chunks per wave on: at 1 M lines -- 3.8 per wave -- one share per wave measured 1716-1743 M lines/s, tickets
        //  1638-1663; at 2 M lines 1804-1844 against 1845-1873, at 16 M 1880 against 2075: profiles/r05v_policy_ab.log)
        prc = head ? fg::plan_launch(fg::k_rfc5424<fg::kWindowKiB, false, true, true>, n, plan_len, 0u, 36864u, sb, &p, *lo,
                                     fg::PlanFormat().classes(2u).lds_tile(fg::sd2::extra_bytes).tile(12288u).chunk(128u).tickets_from(4u))
                   : fg::plan_launch(fg::k_rfc5424<fg::kWindowKiB, false, false, true>, n, plan_len, 0u, 36864u, sb, &p, *lo,
                                     fg::PlanFormat().classes(2u).lds_tile(fg::sd2::extra_bytes).tile(12288u).chunk(128u).tickets_from(4u));
    else
        // (tickets from 20 chunks per wave on: the HBM-bound kernel pays for the first round's burst, fg_pipeline.hpp plan_launch)
        prc = head ? fg::plan_launch(fg::k_rfc5424<fg::kWindowKiB, false, true>, n, plan_len, 0u, 57344u, sb, &p, *lo)
                   : fg::plan_launch(fg::k_rfc5424<fg::kWindowKiB, false>, n, plan_len, 0u, 57344u, sb, &p, *lo, fg::PlanFormat().tickets_from(20u).taper_levels(0u));
    if (prc) return -1;
    if (stash_blocks == 0) stash = nullptr;
    dim3 grid(p.blocks), block(fg::kWave);
    fg::FrameArgs fr{strip, line_bad};
    fg::take_tickets(&fr, tk, p);
    fg::DevTables tt = *t;
    tt.alloc_chunk = fg::entry_chunk(tt.ent_cap, p.blocks, n, *lo, tt.shares);
    unsigned long long* const no_prof = nullptr;
#define FG_LAUNCH_5424(PROF_, HEAD_, SDX_, prof_ptr)                                                                                       \
    hipLaunchKernelGGL((fg::k_rfc5424<fg::kWindowKiB, PROF_, HEAD_, SDX_>), grid, block, p.lds, stream, d_bytes, d_offsets, n, tt, p.tile, p.L, \
                       p.chunk, prof_ptr, stash, fr)
#if defined(FG_PROF_BUILD)
    if (fg::prof_requested()) {
        fg::ProfRun pr;
        if (!pr.begin(stream)) return -1;
        if (sdx) {
            if (head) FG_LAUNCH_5424(true, true, true, pr.d);
            else FG_LAUNCH_5424(true, false, true, pr.d);
        } else {
            if (head) FG_LAUNCH_5424(true, true, false, pr.d);
            else FG_LAUNCH_5424(true, false, false, pr.d);
        }
        pr.end(stream, head ? (sdx ? "rfc5424 (head, pairs)" : "rfc5424 (head)") : (sdx ? "rfc5424 (pairs)" : "rfc5424"), p);
        return (int)hipGetLastError();
    }
#endif
    if (sdx) {
        if (head) FG_LAUNCH_5424(false, true, true, no_prof);
        else FG_LAUNCH_5424(false, false, true, no_prof);
    } else {
        if (head) FG_LAUNCH_5424(false, true, false, no_prof);
        else FG_LAUNCH_5424(false, false, false, no_prof);
    }
#undef FG_LAUNCH_5424
    return (int)hipGetLastError();
}

// (register windows of exactly the tile's rows -- 16 KiB, 12 KiB for the pair-parallel kernel --: the 20 KiB of the plain kernels would
//  hold rows that are never fetched, and the fused loop needs those registers: a spill there is a wait for every load in flight)
// The fused launch (fg_fused.hpp): frame + decode of a raw stream chunk in one kernel.  g from fg::fused_geometry (the caller sized
// `scratch` from it: fg::fused_scratch_bytes).  *d_total = the two device words the launch leaves: lines, abort flag.
extern "C" int fg_launch_rfc5424_fused(const uint8_t* d_bytes, uint64_t nbytes, const fg::DevTables* t, const fg::FusedGeom* g, hipStream_t stream,
                                       uint64_t* stash, uint32_t stash_blocks, uint32_t strip, int final_, uint64_t* d_offsets, uint64_t cap,
                                       uint8_t* scratch, const fg_launch_opts* lo, unsigned long long** d_total) {
    if (nbytes == 0 || !g->ok) return -1;
    const bool sdx = g->variant == 1u;
    const uint32_t base_lds = g->tile + 64u + (g->tile / 16u + 16u) * 2u * (sdx ? 2u : 1u) + (sdx ? fg::sd2::extra_bytes(g->tile) : 0u);
    fg::FusedArgs fa{};
    uint32_t lds = 0, blocks = 0;
    if (stash_blocks == 0) stash = nullptr;
    const uint32_t delim = strip == FG_FRAME_LINE ? 0x0Au : 0u;
    const int prc = sdx ? fg::fused_prepare(fg::k_rfc5424_fused<12, true>, *g, base_lds, nbytes, final_, delim, d_offsets, cap, scratch,
                                            stash ? stash_blocks : 0u, *lo, stream, &fa, &lds, &blocks)
                        : fg::fused_prepare(fg::k_rfc5424_fused<16, false>, *g, base_lds, nbytes, final_, delim, d_offsets, cap, scratch,
                                            stash ? stash_blocks : 0u, *lo, stream, &fa, &lds, &blocks);
    if (prc) return -1;
    fg::DevTables tt = *t;
    tt.alloc_chunk = fg::entry_chunk(tt.ent_cap, blocks, nbytes / (g->S / g->L ? g->S / g->L : 1u) + 1u, *lo, tt.shares);
    *d_total = fa.total;
    if (sdx)
        hipLaunchKernelGGL((fg::k_rfc5424_fused<12, true>), dim3(blocks), dim3(fg::kWave), lds, stream, d_bytes, tt, g->tile, g->L, stash, fa, strip);
    else
        hipLaunchKernelGGL((fg::k_rfc5424_fused<16, false>), dim3(blocks), dim3(fg::kWave), lds, stream, d_bytes, tt, g->tile, g->L, stash, fa, strip);
    return (int)hipGetLastError();
}
