// fg_rfc5424.hip -- gfx950 kernel for RFC5424Decoder::decode
// (reference: src/flowgger/decoder/rfc5424_decoder.rs:17-242).
//
// Work decomposition (HBM-bound byte/integer work, no MFMA):
//   * one 64-lane wavefront owns a group of 64 consecutive lines = one contiguous byte range of
//     the packed buffer;
//   * the range is staged into the wave's LDS tile with coalesced 16-byte-per-lane loads
//     (each HBM byte is fetched exactly once, 1 KiB per wave-instruction);
//   * every lane then tokenises ITS line out of LDS (dword-cached byte reader), so the 64 lines
//     of a group are parsed in parallel by the 64 lanes; a line that does not fit in the tile is
//     parsed by the same code straight from global memory;
//   * fixed-width results go to struct-of-array tables with one coalesced store per column;
//     variable-size structured data goes to an entry table whose slots are reserved with ONE
//     wave-aggregated atomic (count pass -> wave prefix sum -> fill pass, both out of LDS).
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

// scan forward to the next ' ' (or `len`); returns its index
template <class R>
__device__ __forceinline__ uint32_t find_space(R& rd, uint32_t q, uint32_t len) {
    while (q < len && rd.byte(q) != ' ') ++q;
    return q;
}

// parse_data's structured-data walk (rfc5424_decoder.rs:134-158 + parse_sd_data :174-242).
// pos = index of the first '['.  EMIT=false counts entries; EMIT=true writes them starting at
// slot `slot`.  On success *msg_at = index of the ' ' that starts the message.
template <bool EMIT, class R>
__device__ uint32_t sd_walk(R& rd, uint32_t pos, uint32_t len, uint32_t* msg_at, uint32_t* n_ent,
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

// Everything of decode() except writing SD entries.
template <class R>
__device__ void parse_line(R& rd, uint32_t len, Row& r, const DevTables& t) {
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        r.off[k] = 0;
        r.len[k] = FG_NONE;
    }
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
    // parse_data :127-161 on part 7 = [q, len)
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
    // parse_msg :163-172: trim(); "" -> None
    {
        uint32_t s = trim_start(rd, msg_at, len);
        uint32_t e = trim_end(rd, s, len);
        if (e > s) {
            r.off[S_MSG] = s;
            r.len[S_MSG] = e - s;
        }
    }
    // full_msg = line.trim_end() of the BOM-stripped line :46
    {
        uint32_t e = trim_end(rd, line0, len);
        r.off[S_FULL] = line0;
        r.len[S_FULL] = e - line0;
    }
}

// One wave per 64-line group.  Dynamic LDS = the wave's tile (tile_cap bytes, multiple of 16).
__global__ __launch_bounds__(kWave) void k_rfc5424(const uint8_t* __restrict__ bytes,
                                                  const uint64_t* __restrict__ offsets, uint64_t n,
                                                  DevTables t, uint32_t tile_cap) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t lane = threadIdx.x;
    const uint64_t l0 = (uint64_t)blockIdx.x * kWave;
    const uint64_t li = l0 + lane;
    const bool valid = li < n;
    const uint64_t last = (l0 + kWave < n) ? l0 + kWave : n;
    const uint64_t o0 = offsets[valid ? li : last];
    const uint64_t o1 = offsets[valid ? li + 1 : last];
    const uint64_t lo = __shfl(o0, 0, kWave);
    const uint64_t hi = offsets[last];
    const uint64_t a0 = lo & ~15ull;
    uint64_t want = hi - a0;
    const uint32_t span = want > tile_cap ? tile_cap : (uint32_t)((want + 15ull) & ~15ull);

    // stage [a0, a0+span) : 16 B per lane, 1 KiB per wave-instruction, fully coalesced
    {
        const uint4* src = reinterpret_cast<const uint4*>(bytes + a0);
        uint4* dst = reinterpret_cast<uint4*>(smem);
        for (uint32_t k = lane; k < (span >> 4); k += kWave) dst[k] = src[k];
    }
    __syncthreads();

    Row r;
    const uint32_t len = (uint32_t)(o1 - o0);
    const bool in_tile = (o1 - a0) <= (uint64_t)span;
    if (valid) {
        if (in_tile) {
            LdsReader rd(reinterpret_cast<const uint32_t*>(smem), (uint32_t)(o0 - a0));
            parse_line(rd, len, r, t);
        } else {
            GlobalReader rd(reinterpret_cast<const uint32_t*>(bytes), o0);
            parse_line(rd, len, r, t);
        }
    }

    // reserve entry slots: wave prefix sum + one atomic per wave
    uint32_t total;
    uint32_t ex = wave_exclusive_sum(r.n_ent, &total);
    uint32_t first = 0;
    if (total != 0) {  // wave-uniform
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(t.ent_used, (unsigned long long)total);
        base = __shfl(base, 0, kWave);
        unsigned long long mine = base + ex;
        if (r.n_ent != 0) {
            if (mine + r.n_ent > t.ent_cap) {
                r.status = FG_ST_OVERFLOW;
                r.n_ent = 0;
            } else {
                first = (uint32_t)mine;
                uint32_t msg_at, cnt;
                if (in_tile) {
                    LdsReader rd(reinterpret_cast<const uint32_t*>(smem), (uint32_t)(o0 - a0));
                    sd_walk<true>(rd, r.data0, len, &msg_at, &cnt, t, first);
                } else {
                    GlobalReader rd(reinterpret_cast<const uint32_t*>(bytes), o0);
                    sd_walk<true>(rd, r.data0, len, &msg_at, &cnt, t, first);
                }
            }
        }
    }

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

// host-side launcher (called from fg_capi.cpp)
extern "C" int fg_launch_rfc5424(const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t n,
                                 const fg::DevTables* t, uint32_t tile_cap, hipStream_t stream) {
    if (n == 0) return 0;
    uint64_t groups = (n + fg::kWave - 1) / fg::kWave;
    if (groups > 0x7FFFFFFFull) return -1;
    hipLaunchKernelGGL(fg::k_rfc5424, dim3((uint32_t)groups), dim3(fg::kWave), tile_cap, stream, d_bytes, d_offsets, n,
                       *t, tile_cap);
    return (int)hipGetLastError();
}
