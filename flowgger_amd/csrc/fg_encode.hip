// fg_encode.hip -- gfx950 kernels for GelfEncoder::encode FROM THE DECODE TABLES
// (reference: src/flowgger/encoder/gelf_encoder.rs:59-115; serde_json 0.8 `to_vec` of a
//  BTreeMap-backed Value::Object).
//
// The reference builds an ObjectBuilder by inserting, in this order: version, host,
// short_message, timestamp, [level], [full_message], [application_name], [process_id], then for
// every structured-data element [sd_id] and its pairs, then the output.gelf_extra pairs -- a
// later insert replaces an earlier one -- and serialises the BTreeMap: keys in byte order, compact,
// strings through serde_json's `escape_str`, integers through itoa, f64 through the `dtoa` crate
// (Grisu2 + rapidjson's Prettify; fg_dtoa.hpp), non-finite f64 as null.
//
// Here the Record is never materialised: a lane takes the table row of ITS line (spans into the
// packed line bytes + the entry slice) and emits the JSON text directly.
//   * The pairs' keys all start with '_' (every decoder adds it), so the dynamic part is "sort the
//     pair names": 7-byte big-endian prefixes + pair index ranked in registers (BTreeMap order,
//     the later duplicate wins); the static part (the nine fixed keys merged with gelf_extra,
//     conflicts already resolved) is sorted once on the host; the two sorted lists are merged per
//     line with byte comparisons.  More than 32 pairs, or two different names sharing 7 bytes:
//     an exact selection loop (O(n^2) comparisons) instead.
//   * Two passes over the same code with different sinks: size (u32 per line) -> one-workgroup
//     exclusive scan -> write at out[offsets[i]..).  Lines whose decode failed produce nothing.
// v1 scope: records decoded from RFC5424 or LTSV lines (GELF-sourced spans hold JSON escapes that
// would have to be decoded and re-escaped: FG_ERR_UNSUPPORTED for now).  Byte-granular global
// reads / writes: this version is about exactness, not about the roofline.
#include "fg_device.hpp"
#include "fg_dtoa.hpp"

namespace fg {

// static key list entry (host-built, sorted by key): what the value is
enum : uint32_t { SK_APP = 0, SK_FULL = 1, SK_HOST = 2, SK_LEVEL = 3, SK_PROC = 4, SK_SDID = 5, SK_SHORT = 6, SK_TS = 7, SK_VERSION = 8, SK_EXTRA = 9 };
struct StaticKey {
    uint32_t key_off, key_len;  // into the blob
    uint32_t kind;              // SK_*
    uint32_t val_off, val_len;  // SK_EXTRA: the configured value
};
struct EncCfg {
    const uint8_t* blob;       // keys, extra values, LTSV suffixes
    const StaticKey* keys;     // sorted by key bytes
    uint32_t n_keys;
    uint32_t suf_off[4], suf_len[4];  // bool, f64, i64, u64 (len 0xFFFFFFFF = not configured)
    uint32_t src_fmt;
};

struct CountSink {
    uint32_t n = 0;
    __device__ __forceinline__ void put(uint32_t) { ++n; }
};
struct WriteSink {
    uint8_t* p;
    uint32_t n = 0;
    __device__ __forceinline__ void put(uint32_t c) { p[n++] = (uint8_t)c; }
};

constexpr uint32_t kSortSlots = 32;

template <class S>
struct GelfEmitter {
    S& out;
    const EncCfg& cfg;
    GlobalReader rd;           // the line's bytes
    const DevTables& t;
    uint64_t li;
    uint32_t meta;
    bool first_member = true;

    __device__ __forceinline__ void lit(const char* s, uint32_t n) {
        for (uint32_t i = 0; i < n; ++i) out.put((uint32_t)(uint8_t)s[i]);
    }
    __device__ __forceinline__ void esc_byte(uint32_t c) {  // serde_json 0.8 escape_str
        if (c == '"' || c == '\\') {
            out.put('\\');
            out.put(c);
        } else if (c >= 0x20u) {
            out.put(c);
        } else {
            out.put('\\');
            if (c == 8u) out.put('b');
            else if (c == 9u) out.put('t');
            else if (c == 10u) out.put('n');
            else if (c == 12u) out.put('f');
            else if (c == 13u) out.put('r');
            else {
                out.put('u');
                out.put('0');
                out.put('0');
                out.put(c >> 4 ? '1' : '0');
                const uint32_t lo = c & 15u;
                out.put(lo < 10u ? '0' + lo : 'a' + lo - 10u);
            }
        }
    }
    __device__ __forceinline__ void member_start() {
        if (!first_member) out.put(',');
        first_member = false;
    }
    // "key": from the static blob
    __device__ __forceinline__ void key_static(const StaticKey& k) {
        member_start();
        out.put('"');
        for (uint32_t i = 0; i < k.key_len; ++i) esc_byte(cfg.blob[k.key_off + i]);
        out.put('"');
        out.put(':');
    }
    __device__ __forceinline__ void str_span(uint32_t off, uint32_t len) {
        out.put('"');
        for (uint32_t i = 0; i < len; ++i) esc_byte(rd.byte(off + i));
        out.put('"');
    }
    // an RFC5424 SD value with escapes: unescape_sd_value (rfc5424_decoder.rs:105-125), then JSON-escape
    __device__ __forceinline__ void str_span_sd_unescape(uint32_t off, uint32_t len) {
        out.put('"');
        bool esc = false;
        for (uint32_t i = 0; i < len; ++i) {
            const uint32_t c = rd.byte(off + i);
            if (!esc) {
                if (c == '\\') esc = true;
                else esc_byte(c);
            } else {
                if (c != '"' && c != '\\' && c != ']') esc_byte('\\');
                esc_byte(c);
                esc = false;
            }
        }
        out.put('"');
    }
    __device__ __forceinline__ void u64_text(uint64_t v) {
        char buf[20];
        int n = 0;
        do {
            buf[n++] = (char)('0' + (uint32_t)(v % 10u));
            v /= 10u;
        } while (v);
        while (n) out.put((uint32_t)(uint8_t)buf[--n]);
    }
    __device__ __forceinline__ void f64_text(double d) {
        uint64_t b;
        memcpy(&b, &d, 8);
        if (((b >> 52) & 0x7FFu) == 0x7FFu) {  // NaN / inf
            lit("null", 4);
            return;
        }
        char buf[32];
        const int n = dtoa::write(d, buf);
        for (int i = 0; i < n; ++i) out.put((uint32_t)(uint8_t)buf[i]);
    }

    // ---- dynamic keys: entry e's key string is '_' + name [+ LTSV suffix] ---------------------
    struct Dyn {
        uint32_t off, len;      // name span in the line
        uint32_t so, sl;        // suffix in the blob (sl = 0: none)
    };
    __device__ __forceinline__ Dyn dyn_of(uint32_t e) const {
        const fg_span nm = t.ent_name[e];
        Dyn d{nm.off, nm.len, 0u, 0u};
        const uint32_t ty = t.ent_type[e];
        if ((t.ent_flags[e] & FG_EF_SUFFIX) && ty >= FG_T_BOOL && ty <= FG_T_U64 && cfg.suf_len[ty - FG_T_BOOL] != 0xFFFFFFFFu) {
            d.so = cfg.suf_off[ty - FG_T_BOOL];
            d.sl = cfg.suf_len[ty - FG_T_BOOL];
        }
        return d;
    }
    // byte k of the key WITHOUT its leading '_' (k < len + sl)
    __device__ __forceinline__ uint32_t dyn_byte(const Dyn& d, uint32_t k) { return k < d.len ? rd.byte(d.off + k) : cfg.blob[d.so + (k - d.len)]; }
    __device__ __forceinline__ int cmp_dyn(const Dyn& a, const Dyn& b) {
        const uint32_t la = a.len + a.sl, lb = b.len + b.sl, n = la < lb ? la : lb;
        for (uint32_t k = 0; k < n; ++k) {
            const uint32_t x = dyn_byte(a, k), y = dyn_byte(b, k);
            if (x != y) return x < y ? -1 : 1;
        }
        return la == lb ? 0 : (la < lb ? -1 : 1);
    }
    // full key ('_' + ...) against a static key
    __device__ __forceinline__ int cmp_dyn_static(const Dyn& a, const StaticKey& s) {
        if (s.key_len == 0) return 1;
        const uint32_t s0 = cfg.blob[s.key_off];
        if (s0 != '_') return '_' < s0 ? -1 : 1;
        const uint32_t la = a.len + a.sl, lb = s.key_len - 1u, n = la < lb ? la : lb;
        for (uint32_t k = 0; k < n; ++k) {
            const uint32_t x = dyn_byte(a, k), y = cfg.blob[s.key_off + 1u + k];
            if (x != y) return x < y ? -1 : 1;
        }
        return la == lb ? 0 : (la < lb ? -1 : 1);
    }
    __device__ __forceinline__ void emit_dyn(uint32_t e) {
        const Dyn d = dyn_of(e);
        member_start();
        out.put('"');
        out.put('_');
        for (uint32_t k = 0; k < d.len + d.sl; ++k) esc_byte(dyn_byte(d, k));
        out.put('"');
        out.put(':');
        const uint32_t ty = t.ent_type[e];
        const uint64_t v = t.ent_val[e];
        if (ty == FG_T_STRING) {
            const uint32_t vo = (uint32_t)v, vl = (uint32_t)(v >> 32);
            if ((t.ent_flags[e] & FG_EF_VAL_ESC) && cfg.src_fmt == FG_RFC5424) str_span_sd_unescape(vo, vl);
            else str_span(vo, vl);
        } else if (ty == FG_T_BOOL) {
            if (v) lit("true", 4);
            else lit("false", 5);
        } else if (ty == FG_T_NULL) {
            lit("null", 4);
        } else if (ty == FG_T_U64) {
            u64_text(v);
        } else if (ty == FG_T_I64) {
            const int64_t x = (int64_t)v;
            if (x < 0) {
                out.put('-');
                u64_text(0ull - (uint64_t)x);
            } else {
                u64_text((uint64_t)x);
            }
        } else {
            double d2;
            memcpy(&d2, &v, 8);
            f64_text(d2);
        }
    }
    __device__ __forceinline__ void emit_static(const StaticKey& k, uint32_t sdid_entry) {
        const fg_span none{0, FG_NONE};
        fg_span s = none;
        switch (k.kind) {
            case SK_APP: s = t.span[S_APP][li]; if (s.len == FG_NONE) return; break;
            case SK_FULL: s = t.span[S_FULL][li]; if (s.len == FG_NONE) return; break;
            case SK_PROC: s = t.span[S_PROC][li]; if (s.len == FG_NONE) return; break;
            case SK_LEVEL: if (FG_META_SEVERITY(meta) == 0xFFu) return; break;
            case SK_SDID: if (sdid_entry == 0xFFFFFFFFu) return; break;
            default: break;
        }
        key_static(k);
        switch (k.kind) {
            case SK_APP:
            case SK_FULL:
            case SK_PROC: str_span(s.off, s.len); break;
            case SK_HOST: {
                s = t.span[S_HOST][li];
                if (s.len == 0u || s.len == FG_NONE) {
                    out.put('"');
                    lit("unknown", 7);
                    out.put('"');
                } else {
                    str_span(s.off, s.len);
                }
                break;
            }
            case SK_LEVEL: out.put('0' + FG_META_SEVERITY(meta)); break;
            case SK_SDID: {
                const fg_span id = t.ent_name[sdid_entry];
                str_span(id.off, id.len);
                break;
            }
            case SK_SHORT: {
                s = t.span[S_MSG][li];
                if (s.len == FG_NONE) {
                    out.put('"');
                    out.put('-');
                    out.put('"');
                } else {
                    str_span(s.off, s.len);
                }
                break;
            }
            case SK_TS: f64_text(t.ts[li]); break;
            case SK_VERSION:
                out.put('"');
                lit("1.1", 3);
                out.put('"');
                break;
            default: {  // SK_EXTRA
                out.put('"');
                for (uint32_t i = 0; i < k.val_len; ++i) esc_byte(cfg.blob[k.val_off + i]);
                out.put('"');
            }
        }
    }

    // keys64 / slot_ent / order: this lane's LDS scratch (kSortSlots each)
    __device__ __forceinline__ void run(uint64_t* keys64, uint8_t* slot_ent, uint8_t* order) {
        const uint32_t first = t.ent_first[li], cnt = t.ent_count[li];
        // pairs -> slots (and the LAST sd_id: every element's insert replaces the previous one)
        uint32_t sdid_entry = 0xFFFFFFFFu, np = 0;
        bool ranked = cnt <= 255u;
        for (uint32_t e = first; e < first + cnt; ++e) {
            if (t.ent_type[e] == FG_T_SDID) {
                sdid_entry = e;
                continue;
            }
            if (np < kSortSlots && ranked) {
                const Dyn d = dyn_of(e);
                uint64_t pre = 0;
                for (uint32_t k = 0; k < 7u; ++k) pre = (pre << 8) | (k < d.len + d.sl ? dyn_byte(d, k) : 0u);
                keys64[np] = (pre << 8) | np;  // 7 key bytes big-endian, then the slot: equal keys keep insertion order
                slot_ent[np] = (uint8_t)(e - first);
            } else {
                ranked = false;
            }
            ++np;
        }
        if (ranked) {
            uint64_t k[kSortSlots];
#pragma unroll
            for (uint32_t j = 0; j < kSortSlots; ++j) k[j] = j < np ? keys64[j] : ~0ull;
            for (uint32_t i = 0; i < np; ++i) {  // (k[] stays in registers: only the inner loop is unrolled)
                const uint64_t ki = keys64[i];
                uint32_t rank = 0;
#pragma unroll
                for (uint32_t j = 0; j < kSortSlots; ++j) rank += k[j] < ki ? 1u : 0u;
                order[rank] = (uint8_t)i;
            }
            // adjacent equal 7-byte prefixes: duplicates (keep the later insert) or an unresolved order
            for (uint32_t r = 0; r + 1u < np && ranked; ++r) {
                const uint32_t sa = order[r], sb = order[r + 1u];
                if ((keys64[sa] >> 8) != (keys64[sb] >> 8)) continue;
                const int c = cmp_dyn(dyn_of(first + slot_ent[sa]), dyn_of(first + slot_ent[sb]));
                if (c == 0) order[r] = 0xFFu;
                else ranked = false;  // two different names share 7 bytes: exact selection below
            }
        }
        out.put('{');
        uint32_t sk = 0;  // next static key
        if (ranked) {
            for (uint32_t r = 0; r < np; ++r) {
                if (order[r] == 0xFFu) continue;
                const uint32_t e = first + slot_ent[order[r]];
                const Dyn d = dyn_of(e);
                bool shadowed = false;
                while (sk < cfg.n_keys) {
                    const int c = cmp_dyn_static(d, cfg.keys[sk]);
                    if (c < 0) break;
                    if (c == 0) shadowed = true;  // gelf_extra is inserted last: it replaces the pair
                    emit_static(cfg.keys[sk], sdid_entry);
                    ++sk;
                }
                if (!shadowed) emit_dyn(e);
            }
        } else {
            // exact selection: repeatedly the smallest key greater than the previous one; among
            // equal keys the LAST entry (the later insert)
            uint32_t prev = 0xFFFFFFFFu;
            for (;;) {
                uint32_t best = 0xFFFFFFFFu;
                for (uint32_t e = first; e < first + cnt; ++e) {
                    if (t.ent_type[e] == FG_T_SDID) continue;
                    const Dyn d = dyn_of(e);
                    if (prev != 0xFFFFFFFFu && cmp_dyn(d, dyn_of(prev)) <= 0) continue;
                    if (best == 0xFFFFFFFFu || cmp_dyn(d, dyn_of(best)) <= 0) best = e;
                }
                if (best == 0xFFFFFFFFu) break;
                const Dyn d = dyn_of(best);
                bool shadowed = false;
                while (sk < cfg.n_keys) {
                    const int c = cmp_dyn_static(d, cfg.keys[sk]);
                    if (c < 0) break;
                    if (c == 0) shadowed = true;
                    emit_static(cfg.keys[sk], sdid_entry);
                    ++sk;
                }
                if (!shadowed) emit_dyn(best);
                prev = best;
            }
        }
        for (; sk < cfg.n_keys; ++sk) emit_static(cfg.keys[sk], sdid_entry);
        out.put('}');
    }
};

// WRITE = false: sizes[i] = JSON length of line i (0 when its decode failed);
// WRITE = true: the JSON of line i is written at out + out_offsets[i].
template <bool WRITE>
__global__ __launch_bounds__(kWave) void k_gelf_encode(const uint8_t* __restrict__ bytes, const uint64_t* __restrict__ offsets,
                                                      uint64_t n, DevTables t, EncCfg cfg, uint32_t* __restrict__ sizes,
                                                      const uint64_t* __restrict__ out_offsets, uint8_t* __restrict__ out) {
    __shared__ uint64_t s_keys[kWave * kSortSlots];
    __shared__ uint8_t s_slot[kWave * kSortSlots];
    __shared__ uint8_t s_order[kWave * kSortSlots];
    const uint64_t li = (uint64_t)blockIdx.x * kWave + threadIdx.x;
    if (li >= n) return;
    const uint32_t meta = t.meta[li];
    if (FG_META_STATUS(meta) != 0u) {
        if (!WRITE) sizes[li] = 0;
        return;
    }
    GlobalReader rd(reinterpret_cast<const uint32_t*>(bytes), offsets[li]);
    uint64_t* keys64 = s_keys + threadIdx.x * kSortSlots;
    uint8_t* slot_ent = s_slot + threadIdx.x * kSortSlots;
    uint8_t* order = s_order + threadIdx.x * kSortSlots;
    if (WRITE) {
        WriteSink sink{out + out_offsets[li]};
        GelfEmitter<WriteSink> em{sink, cfg, rd, t, li, meta};
        em.run(keys64, slot_ent, order);
    } else {
        CountSink sink;
        GelfEmitter<CountSink> em{sink, cfg, rd, t, li, meta};
        em.run(keys64, slot_ent, order);
        sizes[li] = sink.n;
    }
}

// exclusive scan of sizes[0..n) -> off[0..n], off[n] = total; one workgroup
__global__ __launch_bounds__(1024) void k_sizes_prefix(const uint32_t* __restrict__ sizes, uint64_t n, uint64_t* __restrict__ off) {
    __shared__ uint64_t wave_tot[16];
    __shared__ uint64_t carry_s;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (uint64_t b0 = 0; b0 < n; b0 += 1024) {
        const uint64_t i = b0 + tid;
        const uint64_t x = i < n ? sizes[i] : 0;
        uint64_t inc = x;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            uint64_t y = __shfl_up(inc, d, 64);
            if (lane >= (uint32_t)d) inc += y;
        }
        if (lane == 63) wave_tot[wv] = inc;
        __syncthreads();
        uint64_t o = carry_s;
        for (uint32_t w = 0; w < wv; ++w) o += wave_tot[w];
        if (i < n) off[i] = o + inc - x;
        __syncthreads();
        if (tid == 1023) carry_s = o + inc;
        __syncthreads();
    }
    if (tid == 0) off[n] = carry_s;
}

}  // namespace fg

extern "C" int fg_launch_gelf_encode_sizes(const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t n, const fg::DevTables* t,
                                           const fg::EncCfg* cfg, uint32_t* d_sizes, uint64_t* d_out_offsets, hipStream_t stream) {
    if (n == 0) return 0;
    const uint64_t blocks = (n + fg::kWave - 1) / fg::kWave;
    if (blocks > 0x7FFFFFFFull) return -1;
    hipLaunchKernelGGL(fg::k_gelf_encode<false>, dim3((uint32_t)blocks), dim3(fg::kWave), 0, stream, d_bytes, d_offsets, n, *t, *cfg,
                       d_sizes, (const uint64_t*)nullptr, (uint8_t*)nullptr);
    hipLaunchKernelGGL(fg::k_sizes_prefix, dim3(1), dim3(1024), 0, stream, d_sizes, n, d_out_offsets);
    return (int)hipGetLastError();
}
extern "C" int fg_launch_gelf_encode_write(const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t n, const fg::DevTables* t,
                                           const fg::EncCfg* cfg, const uint64_t* d_out_offsets, uint8_t* d_out, hipStream_t stream) {
    if (n == 0) return 0;
    const uint64_t blocks = (n + fg::kWave - 1) / fg::kWave;
    hipLaunchKernelGGL(fg::k_gelf_encode<true>, dim3((uint32_t)blocks), dim3(fg::kWave), 0, stream, d_bytes, d_offsets, n, *t, *cfg,
                       (uint32_t*)nullptr, d_out_offsets, d_out);
    return (int)hipGetLastError();
}
