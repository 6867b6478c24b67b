// fg_ltsv.hip -- gfx950 kernel for LTSVDecoder::decode
// (reference: src/flowgger/decoder/ltsv_decoder.rs:86-267).
//
// Runs on the streaming pipeline of fg_pipeline.hpp (persistent waves, register prefetch window,
// LDS tile per line group); stage A builds the TAB bitmap.  LTSV is "split on TAB, then on the
// first ':'": every lane walks ITS line part by part -- one bit scan of the TAB bitmap for the
// part's end, one 16-byte window at the part's start for the name, its ':' and the key match --
// instead of byte by byte.  Typed values (input.ltsv_schema) are converted on the GPU with the
// exact Rust semantics of fg_numparse.hpp (f64::from_str is correctly rounded; its rare Decimal
// slow path runs one lane at a time over a per-wave LDS digit buffer); the schema lives in LDS.
// Pairs are parked in the wave's stash while the line is parsed and copied to the entry table
// once the wave has its slots (single parse); the byte-walking two-pass form below stays for
// lines outside the tile and lines with more than kStashEntries pairs.
#include "fg_fused.hpp"
#include "fg_numfold.hpp"
#include "fg_numparse.hpp"
#include "fg_tsfast.hpp"

namespace fg {

enum : uint32_t {
    L_OK = 0,
    L_LEVEL = 1,     // "Invalid severity level"                       :116
    L_LEVEL7 = 2,    // "Severity level should be <= 7"                :118
    L_BOOL = 3,      // "Type error; boolean was expected"             :142
    L_F64 = 4,       // "Type error; f64 was expected"                 :158
    L_I64 = 5,       // "Type error; i64 was expected"                 :174
    L_U64 = 6,       // "Type error; u64 was expected"                 :190
    L_NOTS = 7,      // "Missing timestamp"                            :205
    L_NOHOST = 8,    // "Missing hostname"                             :206
    L_ENGLISH = 9    // "Unable to parse the English to Unix ..."      :252
};

// input.ltsv_schema / input.ltsv_suffixes on the device (ltsv_decoder.rs:24-84)
struct LtsvDevCfg {
    uint32_t n_schema;
    const uint8_t* blob;       // schema names, then the four suffixes, concatenated
    const uint32_t* name_off;  // n_schema + 1 offsets into blob
    const uint8_t* types;      // FG_T_STRING .. FG_T_U64 per name
    uint32_t suf_off[4];       // bool, f64, i64, u64
    uint32_t suf_len[4];
    uint32_t has_suf[4];
};

struct LRow {
    uint32_t status = L_OK;
    uint32_t severity = 0xFF;
    double ts = 0.0;
    uint32_t have_ts = 0, have_host = 0;
    uint32_t host_off = 0, host_len = 0, msg_off = 0, msg_len = FG_NONE;
    uint32_t n_ent = 0;
    uint32_t novalue = 0;  // parts without ':' met so far: the reference println!s "Missing value for name '{}'" for each (:99)
};

// f64::from_str on [b,e); the Decimal slow path is serialised over the wave's LDS digit buffer.
template <class R>
__device__ __forceinline__ bool parse_f64_wave(R& rd, uint32_t b, uint32_t e, uint8_t* lds_digits, double* out) {
    int rc = num::parse_f64(rd, b, e, nullptr, out);
    if (rc != 2) return rc == 1;
    bool pending = true, ok = false;
    while (pending) {
        unsigned long long m = __ballot(pending);
        int leader = __builtin_ctzll(m);
        if ((int)__lane_id() == leader) {
            ok = num::parse_f64(rd, b, e, lds_digits, out) == 1;
            pending = false;
        }
    }
    return ok;
}

// "[day padding:none]/[month repr:short]/[year]:[hour]:[minute]:[second](.[subsecond])?
//  [offset_hour sign:mandatory][offset_minute]"  (ltsv_decoder.rs:236-254)
template <class R>
__device__ bool english_one(R& rd, uint32_t q, uint32_t end, bool with_subsecond, double* out) {
    DateTimeParts p;
    if (q >= end) return false;
    uint32_t d = rd.byte(q) - '0';
    if (d > 9u) return false;
    p.day = (int)d;
    ++q;
    if (q < end) {
        d = rd.byte(q) - '0';
        if (d <= 9u) {
            p.day = p.day * 10 + (int)d;
            ++q;
        }
    }
    if (q >= end || rd.byte(q) != '/') return false;
    ++q;
    if (q + 3 > end) return false;
    {
        // month names, case-sensitive: Jan Feb Mar Apr May Jun Jul Aug Sep Oct Nov Dec
        const uint32_t m = rd.byte(q) | (rd.byte(q + 1) << 8) | (rd.byte(q + 2) << 16);
        const uint32_t names[12] = {0x6E614Au, 0x626546u, 0x72614Du, 0x727041u, 0x79614Du, 0x6E754Au,
                                    0x6C754Au, 0x677541u, 0x706553u, 0x74634Fu, 0x766F4Eu, 0x636544u};
        p.month = 0;
#pragma unroll
        for (int k = 0; k < 12; ++k)
            if (m == names[k]) p.month = k + 1;
        if (!p.month) return false;
        q += 3;
    }
    if (q >= end || rd.byte(q) != '/') return false;
    ++q;
    int ysign = 1;
    if (q < end && (rd.byte(q) == '+' || rd.byte(q) == '-')) {
        ysign = rd.byte(q) == '-' ? -1 : 1;
        ++q;
    }
    if (!take_digits(rd, q, end, 4, &p.year)) return false;
    p.year *= ysign;
    if (q >= end || rd.byte(q) != ':') return false;
    ++q;
    if (!take_digits(rd, q, end, 2, &p.hour)) return false;
    if (q >= end || rd.byte(q) != ':') return false;
    ++q;
    if (!take_digits(rd, q, end, 2, &p.minute)) return false;
    if (q >= end || rd.byte(q) != ':') return false;
    ++q;
    if (!take_digits(rd, q, end, 2, &p.second)) return false;
    p.nano = 0;
    if (with_subsecond) {
        if (q >= end || rd.byte(q) != '.') return false;
        ++q;
        if (!take_subsecond(rd, q, end, &p.nano)) return false;
    }
    if (q >= end || rd.byte(q) != ' ') return false;
    ++q;
    if (q >= end || (rd.byte(q) != '+' && rd.byte(q) != '-')) return false;
    p.off_sign = rd.byte(q) == '-' ? -1 : 1;
    ++q;
    if (!take_digits(rd, q, end, 2, &p.off_h)) return false;
    if (!take_digits(rd, q, end, 2, &p.off_m)) return false;
    if (q != end) return false;
    return datetime_to_unix(p, false, out);
}

// parse_ts (:263-267): Rust f64 -> RFC3339 -> "English"
template <class R>
__device__ bool ltsv_parse_ts(R& rd, uint32_t b, uint32_t e, uint8_t* lds_digits, double* out) {
    if (parse_f64_wave(rd, b, e, lds_digits, out)) return true;
    if (parse_rfc3339(rd, b, e, out)) return true;
    if (english_one(rd, b, e, false, out)) return true;
    return english_one(rd, b, e, true, out);
}

template <class R>
__device__ __forceinline__ bool key_is(R& rd, uint32_t b, uint32_t e, const char* lit, uint32_t n) {
    return num::bytes_equal(rd, b, e, lit, n);
}

// One forward pass over the line.  EMIT=false: validate + count; EMIT=true: write entries.
template <bool EMIT, class R>
__device__ __forceinline__ void ltsv_walk(R& rd, uint32_t len, const LtsvDevCfg& cfg, uint8_t* lds_digits, LRow& r,
                          const DevTables& t, uint32_t slot) {
    uint32_t cnt = 0;
    uint32_t ps = 0;
    for (;;) {  // line.split('\t')
        // find the end of the part and its first ':' in one scan
        uint32_t pe = ps, colon = 0xFFFFFFFFu;
        while (pe < len) {
            uint32_t c = rd.byte(pe);
            if (c == '\t') break;
            if (c == ':' && colon == 0xFFFFFFFFu) colon = pe;
            ++pe;
        }
        if (colon != 0xFFFFFFFFu) {  // else: println!("Missing value for name ...") :99, no effect on the Record
            const uint32_t nb = ps, ne = colon, vb = colon + 1, ve = pe;
            if (key_is(rd, nb, ne, "time", 4)) {
                if (!EMIT) {
                    uint32_t b = vb, e = ve;
                    if (e > b && rd.byte(b) == '[' && rd.byte(e - 1) == ']' && e - b >= 2) {
                        ++b;
                        --e;
                    }
                    double ts;
                    if (!ltsv_parse_ts(rd, b, e, lds_digits, &ts)) {
                        r.status = L_ENGLISH;
                        return;
                    }
                    r.ts = ts;
                    r.have_ts = 1;
                }
            } else if (key_is(rd, nb, ne, "host", 4)) {
                r.host_off = vb;
                r.host_len = ve - vb;
                r.have_host = 1;
            } else if (key_is(rd, nb, ne, "message", 7)) {
                r.msg_off = vb;
                r.msg_len = ve - vb;
            } else if (key_is(rd, nb, ne, "level", 5)) {
                if (!EMIT) {
                    uint64_t lv;
                    if (!num::parse_unsigned(rd, vb, ve, 255, &lv)) {
                        r.status = L_LEVEL;
                        return;
                    }
                    if (lv > 7) {
                        r.status = L_LEVEL7;
                        return;
                    }
                    r.severity = (uint32_t)lv;
                }
            } else {
                // schema lookup (HashMap::get, :126): exact byte match of the name
                uint32_t ty = FG_T_STRING;
                for (uint32_t k = 0; k < cfg.n_schema; ++k) {
                    uint32_t o = cfg.name_off[k], l = cfg.name_off[k + 1] - o;
                    if (l != ne - nb) continue;
                    bool eq = true;
                    for (uint32_t i = 0; i < l && eq; ++i) eq = rd.byte(nb + i) == cfg.blob[o + i];
                    if (eq) {
                        ty = cfg.types[k];
                        break;
                    }
                }
                uint64_t val = (uint64_t)vb | ((uint64_t)(ve - vb) << 32);
                uint32_t flags = 0;
                if (ty != FG_T_STRING) {
                    if (!EMIT || true) {  // the value is needed in both passes (cheap to redo)
                        if (ty == FG_T_BOOL) {
                            if (key_is(rd, vb, ve, "true", 4)) val = 1;
                            else if (key_is(rd, vb, ve, "false", 5)) val = 0;
                            else {
                                r.status = L_BOOL;
                                return;
                            }
                        } else if (ty == FG_T_F64) {
                            double d;
                            if (!parse_f64_wave(rd, vb, ve, lds_digits, &d)) {
                                r.status = L_F64;
                                return;
                            }
                            val = num::f64_to_bits(d);
                        } else if (ty == FG_T_I64) {
                            int64_t x;
                            if (!num::parse_i64(rd, vb, ve, &x)) {
                                r.status = L_I64;
                                return;
                            }
                            val = (uint64_t)x;
                        } else {
                            uint64_t x;
                            if (!num::parse_unsigned(rd, vb, ve, 0xFFFFFFFFFFFFFFFFull, &x)) {
                                r.status = L_U64;
                                return;
                            }
                            val = x;
                        }
                    }
                    // suffix: appended unless the name already ends with it (:131-136)
                    const uint32_t si = ty - FG_T_BOOL;
                    if (cfg.has_suf[si]) {
                        uint32_t sl = cfg.suf_len[si], so = cfg.suf_off[si];
                        bool ends = (ne - nb) >= sl;
                        for (uint32_t i = 0; i < sl && ends; ++i) ends = rd.byte(ne - sl + i) == cfg.blob[so + i];
                        if (!ends) flags |= FG_EF_SUFFIX;
                    }
                }
                if (EMIT) {
                    t.ent_name[slot + cnt] = fg_span{nb, ne - nb};
                    t.ent_val[slot + cnt] = val;
                    t.ent_type[slot + cnt] = (uint8_t)ty;
                    t.ent_flags[slot + cnt] = (uint8_t)flags;
                }
                ++cnt;
            }
        } else if (!EMIT) {
            ++r.novalue;
        }
        if (pe >= len) break;
        ps = pe + 1;
    }
    if (!EMIT) {
        if (!r.have_ts) {
            r.status = L_NOTS;
            return;
        }
        if (!r.have_host) {
            r.status = L_NOHOST;
            return;
        }
        r.n_ent = cnt;
    }
}

// The call the kernels make (rare lines only).  Reader and row travel BY VALUE -- in registers: a reference to either would be an
// object in the caller's frame, i.e. scratch memory reserved for every lane of every wave (508 B per lane in round 3, 140 with the
// row and the reader in the kernel's frame, what is left now is the callees' own).
template <bool EMIT, class R>
__device__ __noinline__ LRow ltsv_walk_call(R rd, uint32_t len, const LtsvDevCfg& cfg, uint8_t* lds_digits, const DevTables& t, uint32_t slot) {
    LRow r;
    ltsv_walk<EMIT>(rd, len, cfg, lds_digits, r, t, slot);
    return r;
}


// ---------------------------------------------------------------------------------------------
// The tile form
// ---------------------------------------------------------------------------------------------
constexpr uint32_t kSchemaLds = 32;    // schema entries mirrored in LDS (more: matched from global memory)
struct SchemaEnt {                      // 24 bytes
    uint32_t len;
    uint32_t type;
    uint32_t name[4];                   // first 16 bytes of the name, zero padded
};
struct SuffixEnt {                      // 16 bytes: the four configured suffixes (bool, f64, i64, u64)
    uint32_t len;                       // 0xFFFFFFFF = not configured
    uint32_t pad;
    uint64_t bytes;                     // first 8 bytes, zero padded
};
constexpr uint32_t kLtsvTablesAt = 768u + kSchemaLds * sizeof(SchemaEnt) + 4u * sizeof(SuffixEnt);  // fg_numfold.hpp tables (8-byte aligned)
constexpr uint32_t kLtsvExtraLds = kLtsvTablesAt + ((numfold::kTableBytes + 15u) & ~15u);

// HEAD = true: the instantiation for LONG lines (persistent_loop<..., HEAD>): only the first kHeadCap bytes of every line are staged and
// tokenised; what lies behind them is only SCANNED for a TAB, wave-cooperatively and without being stored (in practice the long
// part of a log line is its last one, the message) -- a line with a TAB back there, or whose cut part is not a plain string, is
// parsed from global memory.  Three to four times the lines per group on the 64 B .. 8 KiB corpus.
template <bool HEAD = false>
struct LtsvFormatT {
    static constexpr uint32_t kClasses = 1;
    static constexpr int kTailBatch = 16;  // the whole tile beyond the 2 KiB window in one round trip
    // HEAD staging: 512 bytes of every long line (the parts in front of the long one -- in practice the message -- are a few hundred
    // bytes): at the same tile and occupancy a group holds 46 lines of the 64 B .. 8 KiB corpus instead of 28
    static constexpr uint32_t kHeadBytes = 512;
    static constexpr bool kDeferRowStore = false;  // (stage A waits for the tail's loads anyway)
    static __device__ __forceinline__ void classify_store(const uint4& q, uint16_t* bm16, uint32_t chunk, uint32_t, uint32_t) {
        bm16[chunk] = (uint16_t)mask16(q);
    }
    uint32_t n_schema;          // (= cfg->n_schema, in a register)
    const LtsvDevCfg* cfg;      // the kernel's LDS copy: indexed dynamically below, an embedded copy would live in scratch memory
    uint8_t* lds_digits;        // 768-byte digit buffer for dec2flt's slow path
    const SchemaEnt* schema;    // LDS mirror of the first kSchemaLds schema entries
    const SuffixEnt* suffix;    // LDS mirror of the four suffixes
    const double* p10;          // fg_numfold.hpp tables (LDS)
    const uint32_t* dw;
    const DevTables* t_call;    // the tables once more, in LDS: what the byte-wise walk (a real call, rare) is handed -- a reference
                                // to the kernel's own copy would pin that in scratch memory for every line

    // ---- the everyday spellings straight out of registers; false = NOT DECIDED: the byte-wise parser of the grammar runs ----
    // f64::from_str of  -?D+(.D+)?  with a significand below 2^53: Clinger's fast path, w / 10^k is the correctly rounded result
    __device__ __forceinline__ bool fast_f64(const uint32_t w[6], uint32_t n, double* out) const {
        const numfold::Folded f = numfold::fold24(w, n, dw);
        const double d = (double)f.sig / p10[f.nf <= 22u ? f.nf : 0u];
        *out = f.neg ? -d : d;
        return n <= 24u && f.ok && f.sig < (1ull << 53);
    }
    // parse_ts (ltsv_decoder.rs:263-267) of [b, e): the float, RFC3339 and the two English forms; accepted only where the chain
    // f64 -> RFC3339 -> English -> English with subsecond accepts with the same value (an RFC3339 / English text never parses as f64)
    __device__ __forceinline__ bool fast_ts(const Tile& T, uint32_t a, uint32_t n, double* out) const {
        uint32_t r[9];
        {
            const uint32_t d = a >> 2, sh = a & 3u;
            uint32_t x[10];
#pragma unroll
            for (int k = 0; k < 10; ++k) x[k] = T.w[d + k];
#pragma unroll
            for (int k = 0; k < 9; ++k) r[k] = __builtin_amdgcn_alignbyte(x[k + 1], x[k], sh);
        }
        double v0, v1 = 0.0, v2;
        const bool is_f = fast_f64(r, n, &v0);
        const uint32_t c3339 = fast_rfc3339_core(r, n, [&](uint32_t pos, uint32_t* z0, uint32_t* z1) { load8(T, a + pos, z0, z1); }, &v1);
        const bool is_e = fast_english(r, n, &v2);
        *out = is_f ? v0 : c3339 == 1u ? v1 : v2;
        // (c3339 == 2: the byte-wise RFC3339 parser decides, BEFORE the English forms get their turn)
        return is_f || c3339 == 1u || (c3339 == 0u && is_e);
    }

    static __device__ __forceinline__ uint32_t mask16(const uint4& v) { return mask16_eq(v, 0x09090909u); }

    // schema lookup (HashMap::get, ltsv_decoder.rs:126): exact byte match of the name [nb, nb+nl)
    // whose first 16 bytes are in w[] (zero padded)
    __device__ __forceinline__ uint32_t lookup(LdsReader& rd, uint32_t nb, uint32_t nl, const uint32_t w[4]) const {
        const uint32_t ns = n_schema < kSchemaLds ? n_schema : kSchemaLds;
        // four entries per LDS round trip (every lane reads the same entry: a broadcast); the loads of a batch are issued
        // before the first compare
        for (uint32_t k0 = 0; k0 < ns; k0 += 4u) {
            SchemaEnt e[4];
#pragma unroll
            for (uint32_t j = 0; j < 4u; ++j) e[j] = schema[k0 + j < ns ? k0 + j : ns - 1u];
#pragma unroll
            for (uint32_t j = 0; j < 4u; ++j) {
                if (k0 + j >= ns) break;  // scalar
                if (e[j].len == nl && e[j].name[0] == w[0] && e[j].name[1] == w[1] && e[j].name[2] == w[2] && e[j].name[3] == w[3]) {
                    bool eq = true;
                    if (nl > 16u) {
                        const uint32_t o = cfg->name_off[k0 + j];
                        for (uint32_t i = 16; i < nl && eq; ++i) eq = rd.byte(nb + i) == cfg->blob[o + i];
                    }
                    if (eq) return e[j].type;
                }
            }
        }
        for (uint32_t k = kSchemaLds; k < n_schema; ++k) {
            uint32_t o = cfg->name_off[k], l = cfg->name_off[k + 1] - o;
            if (l != nl) continue;
            bool eq = true;
            for (uint32_t i = 0; i < l && eq; ++i) eq = rd.byte(nb + i) == cfg->blob[o + i];
            if (eq) return cfg->types[k];
        }
        return FG_T_STRING;
    }

    // One forward pass over a line in the tile; pairs go to the stash (or, STASH = false, are
    // only counted -- a line with more than kStashEntries pairs is re-walked by ltsv_walk<true>).
    // (measurement build: cycles of  0 part end + name window + ':'  1 time  2 host / message / level  3 schema lookup
    //  4 typed value  5 suffix + stash  6 slots + copy-out; lane 0 adds them up per tile)
    // Where the pairs go while the line is parsed: INTO THE LINE ITSELF.  A lane walks its line left to right; everything before the
    // next part's first byte is dead (names and values are recorded as OFFSETS), so the 16-byte record of a pair is written over the
    // line's own consumed bytes in the tile -- no LDS beyond the tile, no round trip through a stash in global memory (round 2: 96 B
    // per line of HBM traffic and 28 % of stage B).  Records start at the line's first 4-byte boundary; a pair whose record would reach
    // into bytes that are still to be read (a line of many tiny parts: `a:1\tb:2\t...`) clears *in_tile_records, and the line's pairs
    // are written by a second walk over the line in GLOBAL memory (its bytes in the tile are no longer intact).
    // (HEAD: len = the bytes of the line that are in the tile, true_len = its length; rem_clean = no TAB behind the staged bytes;
    //  *redo = the head was not enough: the caller parses the line from global memory)
    template <bool PROF>
    __device__ __forceinline__ void walk_tile(const Tile& T, uint32_t base, uint32_t len, LRow& r, uint32_t* tile_w, bool* in_tile_records,
                                              uint64_t* pc, uint32_t true_len = 0, bool rem_clean = true, bool* redo = nullptr) const {
        uint32_t wpos = (base + 3u) & ~3u;  // tile byte where the next record goes
        bool rec_ok = true;
        uint64_t tk = PROF ? wv::clock() : 0;
        auto tick = [&](int k) {
            if (PROF) {
                const uint64_t now = wv::clock();
                pc[k] += now - tk;
                tk = now;
            }
        };
        LdsReader rd(T.w, base);
        WinReader wr(T, base);  // numbers / timestamps: 16 bytes per LDS round trip
        uint32_t cnt = 0;
        // The TABs of the line out of a 64-bit register window of the bitmap (one LDS round trip covers three or four parts), and the
        // NEXT part's end and name window requested before this part is looked at: the part loop is a chain of dependent LDS
        // round trips, and these two were half of it.
        uint64_t tabw = 0;
        uint32_t tw0 = 0x80000000u;  // line index of the window's first bit (initially: nothing is within 64 of it)
        auto next_tab = [&](uint32_t from) -> uint32_t {  // first TAB at line index >= from, else len
            while (from < len) {
                const uint32_t off = from - tw0;
                if (off < 64u) {
                    const uint64_t m = tabw >> off;
                    if (m) {
                        const uint32_t r = from + (uint32_t)__builtin_ctzll(m);
                        return r < len ? r : len;
                    }
                    from = tw0 + 64u;
                } else {
                    tabw = wv::window64(T.bm, base + from);
                    tw0 = from;
                }
            }
            return len;
        };
        uint32_t ps = 0;
        uint32_t pe = next_tab(0u);
        uint32_t w[4];
        load16(T, base + ps, w);
        for (;;) {  // line.split('\t')
            const bool more = pe < len;
            const bool cutp = HEAD && !more && len < true_len;  // the line's last staged part runs on behind the head
            if (HEAD && cutp && !rem_clean) {
                *redo = true;  // ... and there is a TAB back there: more parts
                return;
            }
            const uint32_t nps = pe + 1u;
            const uint32_t npe = more ? next_tab(nps) : len;
            uint32_t nw[4];
            load16(T, base + (more ? nps : ps), nw);  // (unconditional: a clamped address instead of a branch around the loads)
            // first ':' of the part: in the 16-byte window, else (long name) byte-wise
            const uint32_t plen = pe - ps;
            const uint32_t in_part = plen >= 16u ? 0xFFFFu : (1u << plen) - 1u;
            uint32_t cm = gather16(eq_flags(w[0], 0x3A3A3A3Au), eq_flags(w[1], 0x3A3A3A3Au), eq_flags(w[2], 0x3A3A3A3Au),
                                   eq_flags(w[3], 0x3A3A3A3Au)) & in_part;
            uint32_t colon = 0xFFFFFFFFu;
            if (cm) {
                colon = ps + (uint32_t)__builtin_ctz(cm);
            } else if (plen > 16u) {
                uint32_t q = ps + 16u;
                while (q < pe && rd.byte(q) != ':') ++q;
                if (q < pe) colon = q;
            }
            tick(0);
            if (HEAD && cutp && colon == 0xFFFFFFFFu) {
                *redo = true;  // (its ':' may lie behind the head)
                return;
            }
            if (colon != 0xFFFFFFFFu) {  // else: println!("Missing value for name ...") :99, no effect on the Record
                const uint32_t nb = ps, ne = colon, nl = colon - ps, vb = colon + 1, ve = (HEAD && cutp) ? true_len : pe;
                // the name's first 16 bytes, zero padded (for the key matches)
                uint32_t k[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t lo = 4u * j;
                    k[j] = nl >= lo + 4u ? w[j] : nl <= lo ? 0u : (w[j] & ((1u << (8u * (nl - lo))) - 1u));
                }
                const bool short4 = k[1] == 0u && k[2] == 0u && k[3] == 0u;
                if (nl == 4u && short4 && k[0] == 0x656D6974u) {           // "time"
                    if (HEAD && cutp) {
                        *redo = true;  // (a value that is looked INTO must be there whole)
                        return;
                    }
                    uint32_t b = vb, e = ve;
                    if (e > b && wr.byte(b) == '[' && rd.byte(e - 1) == ']' && e - b >= 2) {
                        ++b;
                        --e;
                    }
                    double ts;
                    if (!fast_ts(T, base + b, e - b, &ts) && !ltsv_parse_ts(wr, b, e, lds_digits, &ts)) {
                        r.status = L_ENGLISH;
                        return;
                    }
                    r.ts = ts;
                    r.have_ts = 1;
                    tick(1);
                } else if (nl == 4u && short4 && k[0] == 0x74736F68u) {    // "host"
                    r.host_off = vb;
                    r.host_len = ve - vb;
                    r.have_host = 1;
                } else if (nl == 7u && k[0] == 0x7373656Du && k[1] == 0x00656761u && k[2] == 0u && k[3] == 0u) {  // "message"
                    r.msg_off = vb;
                    r.msg_len = ve - vb;
                } else if (nl == 5u && k[0] == 0x6576656Cu && k[1] == 0x0000006Cu && k[2] == 0u && k[3] == 0u) {  // "level"
                    if (HEAD && cutp) {
                        *redo = true;
                        return;
                    }
                    uint64_t lv;
                    {
                        uint32_t w6[6];
                        wv::Bytes{T.w}.load24(base + vb, w6);
                        const numfold::Folded f = numfold::fold24(w6, ve - vb, dw);
                        lv = f.sig;
                        if (!(ve - vb <= 24u && f.ok && !f.neg && !f.has_dot && f.sig <= 255u) && !num::parse_unsigned(wr, vb, ve, 255, &lv)) {
                            r.status = L_LEVEL;
                            return;
                        }
                    }
                    if (lv > 7) {
                        r.status = L_LEVEL7;
                        return;
                    }
                    r.severity = (uint32_t)lv;
                    tick(2);
                } else {
                    tick(2);
                    const uint32_t ty = lookup(rd, nb, nl, k);
                    tick(3);
                    uint64_t val = (uint64_t)vb | ((uint64_t)(ve - vb) << 32);
                    uint32_t flags = 0;
                    if (HEAD && cutp && ty != FG_T_STRING) {
                        *redo = true;
                        return;
                    }
                    if (ty != FG_T_STRING) {
                        // the value's first 24 bytes, once; the everyday spellings are decided from them
                        uint32_t w6[6];
                        wv::Bytes{T.w}.load24(base + vb, w6);
                        const uint32_t vn = ve - vb;
                        const numfold::Folded f = numfold::fold24(w6, vn, dw);
                        const bool shaped = vn <= 24u && f.ok;
                        if (ty == FG_T_BOOL) {
                            const bool tt = vn == 4u && w6[0] == 0x65757274u;                            // true
                            const bool ff = vn == 5u && w6[0] == 0x736C6166u && (w6[1] & 0xFFu) == 'e';  // false
                            if (tt) val = 1;
                            else if (ff) val = 0;
                            else {
                                r.status = L_BOOL;
                                return;
                            }
                        } else if (ty == FG_T_F64) {
                            double d = (double)f.sig / p10[f.nf <= 22u ? f.nf : 0u];
                            d = f.neg ? -d : d;
                            if (!(shaped && f.sig < (1ull << 53)) && !parse_f64_wave(wr, vb, ve, lds_digits, &d)) {
                                r.status = L_F64;
                                return;
                            }
                            val = num::f64_to_bits(d);
                        } else if (ty == FG_T_I64) {
                            int64_t x = (int64_t)(f.neg ? 0ull - f.sig : f.sig);
                            const bool fits = f.neg ? f.sig <= (1ull << 63) : f.sig < (1ull << 63);
                            if (!(shaped && !f.has_dot && fits) && !num::parse_i64(wr, vb, ve, &x)) {
                                r.status = L_I64;
                                return;
                            }
                            val = (uint64_t)x;
                        } else {
                            uint64_t x = f.sig;
                            if (!(shaped && !f.has_dot && !f.neg) && !num::parse_unsigned(wr, vb, ve, 0xFFFFFFFFFFFFFFFFull, &x)) {
                                r.status = L_U64;
                                return;
                            }
                            val = x;
                        }
                        tick(4);
                        // suffix: appended unless the name already ends with it (:131-136)
                        const SuffixEnt sf = suffix[ty - FG_T_BOOL];
                        if (sf.len != 0xFFFFFFFFu) {
                            bool ends = nl >= sf.len;
                            if (ends && sf.len != 0u) {
                                if (sf.len <= 8u) {
                                    uint32_t t0, t1;
                                    load8(T, base + ne - sf.len, &t0, &t1);
                                    uint64_t tail = (uint64_t)t0 | ((uint64_t)t1 << 32);
                                    if (sf.len < 8u) tail &= ~0ull >> (64u - 8u * sf.len);
                                    ends = tail == sf.bytes;
                                } else {
                                    const uint32_t so = cfg->suf_off[ty - FG_T_BOOL];
                                    for (uint32_t i = 0; i < sf.len && ends; ++i) ends = rd.byte(ne - sf.len + i) == cfg->blob[so + i];
                                }
                            }
                            if (!ends) flags |= FG_EF_SUFFIX;
                        }
                    }
                    if (rec_ok) {
                        if (wpos + 16u <= base + (more ? nps : len)) {  // (everything before the next part's first byte is dead)
                            uint32_t* d = tile_w + (wpos >> 2);
                            d[0] = nb | (nl << 16);
                            d[1] = ty | (flags << 8);
                            d[2] = (uint32_t)val;
                            d[3] = (uint32_t)(val >> 32);
                            wpos += 16u;
                        } else {
                            rec_ok = false;
                        }
                    }
                    ++cnt;
                    tick(5);
                }
            } else {
                ++r.novalue;
            }
            if (!more) break;
            ps = nps;
            pe = npe;
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = nw[j];
        }
        *in_tile_records = rec_ok;
        if (!r.have_ts) {
            r.status = L_NOTS;
            return;
        }
        if (!r.have_host) {
            r.status = L_NOHOST;
            return;
        }
        r.n_ent = cnt;
    }

    // HEAD: is there a TAB behind the staged head of this lane's line?  All lines of the group at once, wave-cooperatively: the rows
    // of 1 KiB behind the heads are dealt out flat, eight loads in flight, nothing is stored (16-byte aligned: a head ends on a 16-byte
    // boundary of the packed buffer; the buffer descriptor bounds every row at the line's end: lanes beyond fetch zeros).
    __device__ __forceinline__ bool tab_behind_head(const GroupCtx& c, uint32_t len) const {
        const uint32_t lane = threadIdx.x;
        const uint32_t rem = (c.valid && c.tlen != 0u && c.tlen < len) ? len - c.tlen : 0u;
        const uint32_t rows = (rem + 1023u) >> 10;
        uint32_t total;
        const uint32_t first = wv::excl_sum(rows, &total);
        bool has_tab = false;
        const uint64_t rem0 = c.o0 + c.tlen;  // (16-byte aligned when rem != 0)
        for (uint32_t r0 = 0; r0 < total; r0 += 8u) {
            u32x4 v[8];
            uint32_t owner[8], left[8];
#pragma unroll
            for (uint32_t j = 0; j < 8u; ++j) {
                const uint32_t row = r0 + j < total ? r0 + j : total - 1u;
                // the line that owns the row: the last lane whose first row is <= row (rows are dealt out in lane order)
                const unsigned long long m = __ballot(rows != 0u && first <= row);
                const uint32_t k = 63u - (uint32_t)__builtin_clzll(m);
                owner[j] = k;
                const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)rem0, (int)k);
                const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(rem0 >> 32), (int)k);
                const uint32_t kfirst = (uint32_t)__builtin_amdgcn_readlane((int)first, (int)k);
                const uint32_t krem = (uint32_t)__builtin_amdgcn_readlane((int)rem, (int)k);
                const uint32_t off = (row - kfirst) << 10;
                const uint64_t a = ((uint64_t)lo | ((uint64_t)hi << 32)) + off;
                // (the range in whole 16-byte chunks -- a 16-byte load that straddles the end of the range fetches nothing; the packed
                //  buffer is readable to its size rounded up to 16 -- and the bytes behind the line's end masked out below)
                left[j] = krem - off;
                __amdgpu_buffer_rsrc_t rsrc =
                    __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(c.bytes + a), (short)0, (int)((left[j] + 15u) & ~15u), 0x00020000);
                v[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(lane * 16u), 0, FG_STREAM_AUX);
            }
#pragma unroll
            for (uint32_t j = 0; j < 8u; ++j) {
                uint32_t hit = 0;
#pragma unroll
                for (uint32_t d = 0; d < 4u; ++d) {
                    const uint32_t at = lane * 16u + 4u * d;  // this dword's first byte in the row
                    const uint32_t nb = left[j] > at ? (left[j] - at < 4u ? left[j] - at : 4u) : 0u;
                    const uint32_t keep = nb == 4u ? 0xFFFFFFFFu : (1u << (8u * nb)) - 1u;
                    hit |= eq_flags(v[j][d], 0x09090909u) & keep;
                }
                const bool any_tab = __ballot(hit != 0u) != 0ull;  // wave-uniform
                if (any_tab && lane == owner[j]) has_tab = true;
            }
        }
        return has_tab;
    }

    __device__ __forceinline__ RowOut decode(const GroupCtx& c, const DevTables& t) const {
        const uint32_t lane = threadIdx.x;
        const uint32_t len = (uint32_t)(c.o1 - c.o0);
        // HEAD staging: the tile holds the line's first c.tlen bytes at c.tbase -- the whole line, or its head only
        const bool whole = HEAD ? c.tlen == len : (c.o1 - c.a0) <= (uint64_t)c.span;
        const bool head_only = HEAD && !whole && c.tlen >= 256u;
        const bool in_tile = whole || head_only;  // the tile walk can start; (head_only: it may hand the line back)
        const uint32_t base = HEAD ? c.tbase : (uint32_t)(c.o0 - c.a0);
        bool rem_clean = true;
        if constexpr (HEAD) rem_clean = !tab_behind_head(c, len);
        const uint32_t slen = HEAD ? c.tlen : len;  // bytes of the line the tile walk may look at
        Tile T{reinterpret_cast<const uint32_t*>(c.smem), reinterpret_cast<const uint32_t*>(c.bm16)};
        LRow r;
        const bool name_fits = len < 65536u;  // the in-tile records keep 16-bit name offsets
        uint32_t* tile_w = reinterpret_cast<uint32_t*>(const_cast<uint8_t*>(c.smem));
        bool in_tile_records = false;
        bool tile_lane = c.valid && in_tile && name_fits;
        uint64_t pc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        uint64_t t6 = c.phase ? wv::clock() : 0;
        if (c.valid) {
            if (tile_lane) {
                bool redo = false;
                if (c.phase) walk_tile<true>(T, base, slen, r, tile_w, &in_tile_records, pc, len, rem_clean, &redo);
                else walk_tile<false>(T, base, slen, r, tile_w, &in_tile_records, pc, len, rem_clean, &redo);
                if (c.phase) t6 = wv::clock();
                if (HEAD && redo) {  // the head was not enough: the whole line, from global memory (below)
                    tile_lane = false;
                    r = LRow();
                }
            }
            if (!tile_lane) {
                // (rare; a real call: its LRow lives in memory, so it gets its own -- `r` must never have its address taken, or
                //  the fast path's row would live in scratch memory too)
                // The same goes for the tables and the configuration: passed by reference from HERE they would be kept in scratch
                // memory for the whole kernel (every table store then reloads its column pointer from there): the call gets the
                // copies the kernel parked in LDS.
                const DevTables& t_copy = *t_call;
                const LtsvDevCfg& cfg_copy = *cfg;
                if (whole && !HEAD) r = ltsv_walk_call<false>(LdsReader(T.w, base), len, cfg_copy, lds_digits, t_copy, 0);
                else r = ltsv_walk_call<false>(GlobalReader(reinterpret_cast<const uint32_t*>(c.bytes), c.o0), len, cfg_copy, lds_digits, t_copy, 0);
            }
            if (r.status != L_OK) r.n_ent = 0;
        }
        const EntAlloc ea = alloc_entries_ex(t, r.n_ent, c.ent_state);
        if (ea.overflow) {
            r.status = FG_ST_OVERFLOW;
            r.n_ent = 0;
        }
        const uint32_t first = (ea.overflow || ea.total == 0u) ? 0u : ea.s.at(ea.ex);
        if (r.n_ent != 0) {
            if (tile_lane && in_tile_records) {
                // the records sit in the line's own (consumed) bytes: four in flight before the first store (the compiler cannot hoist
                // an LDS read above a table store: for all it knows they alias)
                const uint32_t* rec = tile_w + (((base + 3u) & ~3u) >> 2);
                for (uint32_t k0 = 0; k0 < r.n_ent; k0 += 4u) {
                    uint32_t q[4][4];
#pragma unroll
                    for (uint32_t j = 0; j < 4u; ++j) {
                        const uint32_t k = k0 + j < r.n_ent ? k0 + j : r.n_ent - 1u;
#pragma unroll
                        for (uint32_t d = 0; d < 4u; ++d) q[j][d] = rec[k * 4u + d];
                    }
#pragma unroll
                    for (uint32_t j = 0; j < 4u; ++j) {
                        const uint32_t k = k0 + j;
                        if (k < r.n_ent) {
                            gstore(t.ent_name, first + k, fg_span{q[j][0] & 0xFFFFu, q[j][0] >> 16});
                            gstore(t.ent_val, first + k, (uint64_t)q[j][2] | ((uint64_t)q[j][3] << 32));
                            gstore(t.ent_type, first + k, (uint8_t)(q[j][1] & 0xFFu));
                            gstore(t.ent_flags, first + k, (uint8_t)((q[j][1] >> 8) & 0xFFu));
                        }
                    }
                }
            } else {
                // (a line outside the tile -- or one whose records did not fit its consumed bytes, whose copy in the tile is therefore
                //  no longer intact: from global memory then)
                const DevTables& t_copy = *t_call;  // (see above)
                const LtsvDevCfg& cfg_copy = *cfg;
                if (whole && !HEAD && !tile_lane) (void)ltsv_walk_call<true>(LdsReader(T.w, base), len, cfg_copy, lds_digits, t_copy, first);
                else (void)ltsv_walk_call<true>(GlobalReader(reinterpret_cast<const uint32_t*>(c.bytes), c.o0), len, cfg_copy, lds_digits, t_copy, first);
            }
        }
        if (c.phase) {
            pc[6] += wv::clock() - t6;
            if (lane == 0)
                for (int k = 0; k < 7; ++k) atomicAdd(c.phase + k, (unsigned long long)pc[k]);
        }
        RowOut o;
        const bool ok = r.status == L_OK;
        const fg_span none{0, FG_NONE};
        // FG_F_LTSV_NOVALUE: the reference wrote to stdout while it decoded this line; a row whose decode FAILED says in
        // hostname.off how many parts it had printed for by then (the walk stops at the first error, :116-190)
        // (and, saturated at 254, in the meta word's facility byte -- always None for LTSV otherwise -- for callers that only get the
        //  meta column back: fg_transcode_batch)
        const uint32_t fac = (!ok && r.novalue) ? (r.novalue < 254u ? r.novalue : 254u) : 0xFFu;
        o.meta = r.status | (fac << 8) | ((ok ? r.severity : 0xFFu) << 16) | ((r.novalue ? (uint32_t)FG_F_LTSV_NOVALUE : 0u) << 24);
        o.ts = ok ? r.ts : 0.0;
        o.span[S_HOST] = ok ? fg_span{r.host_off, r.host_len} : fg_span{r.novalue, FG_NONE};
        o.span[S_APP] = none;
        o.span[S_PROC] = none;
        o.span[S_MSGID] = none;
        o.span[S_MSG] = ok ? fg_span{r.msg_off, r.msg_len} : none;
        o.span[S_FULL] = ok ? fg_span{0, len} : none;  // full_msg = Some(line), untrimmed :218
        o.first = first;
        o.count = r.n_ent;
        return o;
    }
};

using LtsvFormat = LtsvFormatT<false>;

// The per-wave setup both streaming skeletons share (schema / suffix mirrors, the digit-fold tables, the LDS copies of the tables and
// the configuration), then `run(fmt, tables)`.
template <bool HEAD, class Run>
__device__ __forceinline__ void ltsv_body(const DevTables& t, const LtsvDevCfg& cfg, uint32_t tile_cap, Run run) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t* extra = smem + tile_cap + 64u + (tile_cap / 16u + 16u) * 2u;
    SchemaEnt* schema = reinterpret_cast<SchemaEnt*>(extra + 768u);
    // mirror the schema into LDS once per wave (names zero padded to 16 bytes)
    const uint32_t ns = cfg.n_schema < kSchemaLds ? cfg.n_schema : kSchemaLds;
    for (uint32_t k = threadIdx.x; k < ns; k += kWave) {
        const uint32_t o = cfg.name_off[k], l = cfg.name_off[k + 1] - o;
        SchemaEnt e;
        e.len = l;
        e.type = cfg.types[k];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint32_t v = 0;
            for (uint32_t i = 0; i < 4; ++i) {
                const uint32_t idx = 4u * j + i;
                if (idx < l) v |= (uint32_t)cfg.blob[o + idx] << (8u * i);
            }
            e.name[j] = v;
        }
        schema[k] = e;
    }
    SuffixEnt* suffix = reinterpret_cast<SuffixEnt*>(extra + 768u + kSchemaLds * sizeof(SchemaEnt));
    if (threadIdx.x < 4u) {
        const uint32_t k = threadIdx.x;
        SuffixEnt e;
        e.len = cfg.has_suf[k] ? cfg.suf_len[k] : 0xFFFFFFFFu;
        e.pad = 0;
        e.bytes = 0;
        if (cfg.has_suf[k])
            for (uint32_t i = 0; i < cfg.suf_len[k] && i < 8u; ++i) e.bytes |= (uint64_t)cfg.blob[cfg.suf_off[k] + i] << (8u * i);
        suffix[k] = e;
    }
    double* p10 = reinterpret_cast<double*>(extra + kLtsvTablesAt);
    uint32_t* dw = reinterpret_cast<uint32_t*>(extra + kLtsvTablesAt + numfold::kP10Words * 8u);
    numfold::init_tables(dw, p10);
    __shared__ DevTables t_call;
    __shared__ LtsvDevCfg cfg_call;
    if (threadIdx.x == 0) {
        t_call = t;
        cfg_call = cfg;
    }
    __syncthreads();
    LtsvFormatT<HEAD> fmt{cfg.n_schema, &cfg_call, extra, schema, suffix, p10, dw, &t_call};
    run(fmt, t_call);
}

template <int NB, bool PROF, bool HEAD = false>
__global__ __launch_bounds__(kWave, 2) void k_ltsv(const uint8_t* __restrict__ bytes, const uint64_t* __restrict__ offsets,
                                                  uint64_t n, DevTables t, LtsvDevCfg cfg, uint32_t tile_cap, uint32_t L,
                                                  uint64_t groups, unsigned long long* prof, uint64_t* stash_base, FrameArgs fr) {
    // (the pipeline gets the LDS copy of the tables too: see k_gelf -- scalar-register tuples parked in VGPR lanes otherwise)
    ltsv_body<HEAD>(t, cfg, tile_cap, [&](LtsvFormatT<HEAD>& fmt, DevTables& t_call) {
        persistent_loop<NB, PROF, LtsvFormatT<HEAD>, HEAD>(bytes, offsets, n, t_call, tile_cap, L, groups, prof, stash_base, fmt, fr);
    });
}

// The same decoder over a RAW stream: the kernel frames its tiles itself (fg_fused.hpp)
template <int NB>
__global__ __launch_bounds__(kWave, 2) void k_ltsv_fused(const uint8_t* __restrict__ bytes, DevTables t, LtsvDevCfg cfg, uint32_t tile_cap,
                                                        uint32_t L, uint64_t* stash_base, FusedArgs fa, uint32_t strip) {
    __shared__ FusedArgs fa_lds;  // (see k_rfc5424_fused)
    if (threadIdx.x == 0) fa_lds = fa;
    ltsv_body<false>(t, cfg, tile_cap, [&](LtsvFormatT<false>& fmt, DevTables& t_call) {
        fused_loop<NB>(bytes, t_call, tile_cap, L, fmt, fa_lds, strip, stash_base);
    });
}

}  // namespace fg

extern "C" int fg_launch_ltsv(const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t n, const fg::DevTables* t,
                              const fg::LtsvDevCfg* cfg, uint64_t avg_len, hipStream_t stream, uint64_t* stash,
                              uint32_t stash_blocks, uint32_t strip, const uint8_t* line_bad, const fg_launch_opts* lo, fg::TicketSlot* tk) {
    if (n == 0) return 0;
    fg::LaunchPlan p;
    // long lines: only the head of every line is staged, the rest is scanned for a TAB (LtsvFormatT<true>)
    const bool head = (lo->flags & FG_LO_FORCE_HEAD) || (avg_len >= 768u && !(lo->flags & FG_LO_NO_HEAD));
    const uint64_t plan_len = head ? (avg_len < fg::LtsvFormatT<true>::kHeadBytes ? avg_len : fg::LtsvFormatT<true>::kHeadBytes) : avg_len;
    if (head ? fg::plan_launch(fg::k_ltsv<fg::kComputeBoundWindow, false, true>, n, plan_len, fg::kLtsvExtraLds, 57344u, stash ? stash_blocks : 0u, &p, *lo)
             // (whole lines staged: chunks of 128 lines -- 3.78 vs 3.67 G lines/s with the pipeline's 256, alternated on one box,
             //  profiles/r04z3_sweep_ltsv.log)
             : fg::plan_launch(fg::k_ltsv<fg::kComputeBoundWindow, false>, n, plan_len, fg::kLtsvExtraLds, 57344u, stash ? stash_blocks : 0u, &p, *lo,
                               fg::PlanFormat().chunk(128u)))
        return -1;
    if (stash_blocks == 0) stash = nullptr;
    dim3 grid(p.blocks), block(fg::kWave);
    fg::FrameArgs fr{strip, line_bad};
    fg::take_tickets(&fr, tk, p);
    fg::DevTables tt = *t;
    tt.alloc_chunk = fg::entry_chunk(tt.ent_cap, p.blocks, n, *lo, tt.shares);
#if defined(FG_PROF_BUILD)
    if (fg::prof_requested()) {
        fg::ProfRun pr;
        if (!pr.begin(stream)) return -1;
        if (head)
            hipLaunchKernelGGL((fg::k_ltsv<fg::kComputeBoundWindow, true, true>), grid, block, p.lds, stream, d_bytes, d_offsets, n, tt, *cfg, p.tile,
                               p.L, p.chunk, pr.d, stash, fr);
        else
            hipLaunchKernelGGL((fg::k_ltsv<fg::kComputeBoundWindow, true>), grid, block, p.lds, stream, d_bytes, d_offsets, n, tt, *cfg, p.tile,
                               p.L, p.chunk, pr.d, stash, fr);
        pr.end(stream, head ? "ltsv (head)" : "ltsv", p);
        return (int)hipGetLastError();
    }
#endif
    if (head)
        hipLaunchKernelGGL((fg::k_ltsv<fg::kComputeBoundWindow, false, true>), grid, block, p.lds, stream, d_bytes, d_offsets, n, tt, *cfg, p.tile,
                           p.L, p.chunk, (unsigned long long*)nullptr, stash, fr);
    else
        hipLaunchKernelGGL((fg::k_ltsv<fg::kComputeBoundWindow, false>), grid, block, p.lds, stream, d_bytes, d_offsets, n, tt, *cfg, p.tile,
                           p.L, p.chunk, (unsigned long long*)nullptr, stash, fr);
    return (int)hipGetLastError();
}

// The fused launch (fg_fused.hpp): frame + decode of a raw stream chunk in one kernel (see fg_launch_rfc5424_fused).
extern "C" int fg_launch_ltsv_fused(const uint8_t* d_bytes, uint64_t nbytes, const fg::DevTables* t, const fg::LtsvDevCfg* cfg, const fg::FusedGeom* g,
                                    hipStream_t stream, uint64_t* stash, uint32_t stash_blocks, uint32_t strip, int final_, uint64_t* d_offsets,
                                    uint64_t cap, uint8_t* scratch, const fg_launch_opts* lo, unsigned long long** d_total) {
    if (nbytes == 0 || !g->ok) return -1;
    const uint32_t base_lds = g->tile + 64u + (g->tile / 16u + 16u) * 2u + fg::kLtsvExtraLds;
    fg::FusedArgs fa{};
    uint32_t lds = 0, blocks = 0;
    if (stash_blocks == 0) stash = nullptr;
    const uint32_t delim = strip == FG_FRAME_LINE ? 0x0Au : 0u;
    if (fg::fused_prepare(fg::k_ltsv_fused<fg::kComputeBoundWindow>, *g, base_lds, nbytes, final_, delim, d_offsets, cap, scratch, stash ? stash_blocks : 0u, *lo,
                          stream, &fa, &lds, &blocks))
        return -1;
    fg::DevTables tt = *t;
    tt.alloc_chunk = fg::entry_chunk(tt.ent_cap, blocks, nbytes / (g->S / g->L ? g->S / g->L : 1u) + 1u, *lo, tt.shares);
    *d_total = fa.total;
    hipLaunchKernelGGL((fg::k_ltsv_fused<fg::kComputeBoundWindow>), dim3(blocks), dim3(fg::kWave), lds, stream, d_bytes, tt, *cfg, g->tile, g->L, stash, fa, strip);
    return (int)hipGetLastError();
}
