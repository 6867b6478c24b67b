// fg_rfc3164.hip -- gfx950 kernel for RFC3164Decoder::decode (rfc3164_decoder.rs:31-213).
//
// The per-line logic is fg_rfc3164_parse.hpp (host + device; checked on the CPU against the oracle).  Launch
// geometry: one 64-lane workgroup = 64 consecutive lines = ONE contiguous byte range of the packed buffer, staged
// into LDS with coalesced 16-byte loads; every lane tokenises ITS line out of LDS (whitespace tokens, month / time
// literals, zone-name lookup in the sorted table, calendar arithmetic) and stores the 68-byte table row (struct of
// arrays: the stores of a wave are coalesced per column).  A group whose bytes exceed the tile (long lines) reads
// from global memory.  RFC3164 produces no structured data: no entry table traffic.
// Roofline: HBM -- line bytes + 8 B offset read once, 68 B written per line.
// (Round 5, measured and dropped: this kernel as a format of the shared streaming pipeline -- persistent waves, the next group
//  prefetched into the register window, chunks by ticket: 3.70 against 3.84 G lines/s at 100 M lines, 47 / 49 / 96 / 357 us against
//  46 / 48 / 101 / 311 us at 16 K .. 1 M lines, profiles/r05ab_rfc3164_pipeline_ab.log.  A group takes ~46 us however it is staged:
//  the time is the lane-serial parser -- three line shapes in one wave, zone names resolved through dependent loads from global
//  memory -- and the one-workgroup form keeps nine to ten such waves on a CU where the pipeline's window leaves seven or eight.)
#include "fg_device.hpp"
#include "fg_rfc3164_parse.hpp"

namespace fg {

struct R3164Args {
    r3164::Cfg cfg;
    uint32_t strip;           // FG_FRAME_NONE / _LINE / _NUL: terminators to drop from the frame
    const uint8_t* line_bad;  // [n] 1 = not valid UTF-8 (or null)
};

template <class R>
__device__ __forceinline__ void r3164_lane(R rd, uint32_t len, uint64_t li, const DevTables& t, const R3164Args& a) {
    // terminator stripping (BufRead::lines / split(0) semantics)
    if (a.strip != FG_FRAME_NONE && len) {
        const uint32_t b1 = rd.byte(len - 1);
        if (a.strip == FG_FRAME_LINE) {
            if (b1 == '\n') {
                --len;
                if (len && rd.byte(len - 1) == '\r') --len;
            }
        } else if (b1 == 0u) {
            --len;
        }
    }
    r3164::Row r;
    if (a.line_bad && a.line_bad[li]) {
        r.status = FG_ST_BAD_UTF8;
    } else {
        r3164::parse_line(rd, len, a.cfg, r);
    }
    const bool ok = r.status == r3164::ST_OK;
    const fg_span none{0u, FG_NONE};
    t.meta[li] = r.status | (ok ? r.fac : 0xFFu) << 8 | (ok ? r.sev : 0xFFu) << 16 | (ok && r.msg_join ? (uint32_t)FG_F_MSG_JOIN : 0u) << 24;
    t.ts[li] = ok ? r.ts : 0.0;
    t.span[S_HOST][li] = ok ? fg_span{r.host_off, r.host_len} : none;
    t.span[S_APP][li] = none;
    t.span[S_PROC][li] = none;
    t.span[S_MSGID][li] = none;
    t.span[S_MSG][li] = ok ? fg_span{r.msg_off, r.msg_len} : none;
    t.span[S_FULL][li] = ok ? fg_span{0u, r.full_len} : none;
    t.ent_first[li] = 0;
    t.ent_count[li] = 0;
}

__global__ __launch_bounds__(kWave) void k_rfc3164(const uint8_t* __restrict__ bytes, const uint64_t* __restrict__ offsets, uint64_t n,
                                                  DevTables t, R3164Args a, uint32_t tile_cap) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_tile[];  // tile_cap + 16 bytes
    const uint64_t g0 = (uint64_t)blockIdx.x * kWave;
    const uint64_t g1 = g0 + kWave < n ? g0 + kWave : n;
    const uint64_t li = g0 + threadIdx.x;
    // the lane's own offsets first (in flight while the tile is staged; fetched after it they are a dependent round trip)
    const bool live = li < n;
    const uint64_t lic = live ? li : n - 1u;
    const uint64_t o0 = offsets[lic], o1 = offsets[lic + 1u];
    const uint64_t a_begin = offsets[g0], a_end = offsets[g1];
    const uint64_t a0 = a_begin & ~15ull;
    const uint64_t span = (a_end - a0 + 15ull) & ~15ull;
    if (span <= (uint64_t)tile_cap) {
        stage_tile<20>(bytes, a0, (uint32_t)span, s_tile);  // (20 KiB per round trip: the usual tile in one)
        __syncthreads();  // single-wave workgroup: orders the LDS writes before the lanes' reads
        if (live) {
            LdsReader rd(reinterpret_cast<const uint32_t*>(s_tile), (uint32_t)(o0 - a0));
            r3164_lane(rd, (uint32_t)(o1 - o0), li, t, a);
        }
    } else if (live) {
        GlobalReader rd(reinterpret_cast<const uint32_t*>(bytes), o0);
        r3164_lane(rd, (uint32_t)(o1 - o0), li, t, a);
    }
}

}  // namespace fg

extern "C" int fg_launch_rfc3164(const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t n, const fg::DevTables* t,
                                 const fg::r3164::Cfg* cfg, uint32_t tile_cap, hipStream_t stream, uint32_t strip,
                                 const uint8_t* line_bad) {
    if (n == 0) return 0;
    const uint64_t blocks = (n + fg::kWave - 1) / fg::kWave;
    if (blocks > 0x7FFFFFFFull) return -1;
    fg::R3164Args a{*cfg, strip, line_bad};
    hipLaunchKernelGGL(fg::k_rfc3164, dim3((uint32_t)blocks), dim3(fg::kWave), tile_cap + 16u, stream, d_bytes, d_offsets, n, *t, a, tile_cap);
    return (int)hipGetLastError();
}
