// fg_ltsv.hip -- gfx950 kernel for LTSVDecoder::decode
// (reference: src/flowgger/decoder/ltsv_decoder.rs:86-267).
//
// Same decomposition as the RFC5424 kernel: one wave per 64 lines, the group's byte range is
// streamed HBM -> LDS with coalesced 16 B/lane loads, then every lane walks ITS line out of LDS.
// LTSV is "split on TAB, then on the first ':'", so the walk is a single forward pass; typed
// values (input.ltsv_schema) are converted on the GPU with the exact Rust semantics of
// fg_numparse.hpp (f64::from_str is correctly rounded; its rare Decimal slow path runs one lane
// at a time over a per-wave LDS digit buffer).
// Pairs go to the shared entry table: count pass -> one wave-aggregated atomic -> fill pass.
#include "fg_device.hpp"
#include "fg_numparse.hpp"

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
__device__ void ltsv_walk(R& rd, uint32_t len, const LtsvDevCfg& cfg, uint8_t* lds_digits, LRow& r,
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

__global__ __launch_bounds__(kWave) void k_ltsv(const uint8_t* __restrict__ bytes, const uint64_t* __restrict__ offsets,
                                               uint64_t n, DevTables t, LtsvDevCfg cfg, uint32_t tile_cap) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t* lds_digits = smem + tile_cap + 64u;  // 768-byte digit buffer for dec2flt's slow path
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
    stage_tile(bytes, a0, span, smem);
    __syncthreads();

    LRow r;
    const uint32_t len = (uint32_t)(o1 - o0);
    const bool in_tile = (o1 - a0) <= (uint64_t)span;
    const uint32_t base = (uint32_t)(o0 - a0);
    if (valid) {
        if (in_tile) {
            LdsReader rd(reinterpret_cast<const uint32_t*>(smem), base);
            ltsv_walk<false>(rd, len, cfg, lds_digits, r, t, 0);
        } else {
            GlobalReader rd(reinterpret_cast<const uint32_t*>(bytes), o0);
            ltsv_walk<false>(rd, len, cfg, lds_digits, r, t, 0);
        }
        if (r.status != L_OK) r.n_ent = 0;
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
                LRow scratch = r;
                if (in_tile) {
                    LdsReader rd(reinterpret_cast<const uint32_t*>(smem), base);
                    ltsv_walk<true>(rd, len, cfg, lds_digits, scratch, t, first);
                } else {
                    GlobalReader rd(reinterpret_cast<const uint32_t*>(bytes), o0);
                    ltsv_walk<true>(rd, len, cfg, lds_digits, scratch, t, first);
                }
            }
        }
    }
    if (valid) {
        const bool ok = r.status == L_OK;
        const fg_span none{0, FG_NONE};
        t.meta[li] = r.status | (0xFFu << 8) | ((ok ? r.severity : 0xFFu) << 16);
        t.ts[li] = ok ? r.ts : 0.0;
        t.span[S_HOST][li] = ok ? fg_span{r.host_off, r.host_len} : none;
        t.span[S_APP][li] = none;
        t.span[S_PROC][li] = none;
        t.span[S_MSGID][li] = none;
        t.span[S_MSG][li] = ok ? fg_span{r.msg_off, r.msg_len} : none;
        t.span[S_FULL][li] = ok ? fg_span{0, len} : none;  // full_msg = Some(line), untrimmed :218
        t.ent_first[li] = first;
        t.ent_count[li] = r.n_ent;
    }
}

}  // namespace fg

extern "C" int fg_launch_ltsv(const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t n, const fg::DevTables* t,
                              const fg::LtsvDevCfg* cfg, uint32_t tile_cap, hipStream_t stream) {
    if (n == 0) return 0;
    uint64_t groups = (n + fg::kWave - 1) / fg::kWave;
    if (groups > 0x7FFFFFFFull) return -1;
    uint32_t lds = tile_cap + 64u + 768u;
    hipLaunchKernelGGL(fg::k_ltsv, dim3((uint32_t)groups), dim3(fg::kWave), lds, stream, d_bytes, d_offsets, n, *t, *cfg,
                       tile_cap);
    return (int)hipGetLastError();
}
