// fg_encode.hip -- gfx950 kernels for Encoder::encode + Merger::frame FROM THE DECODE TABLES:
// GELF, LTSV, RFC5424, RFC3164 and passthrough encoders, line / nul / syslen mergers, for records decoded by any of
// the three decoders.  The per-record logic is fg_emit.hpp (host-tested against the oracle); this file is the launch
// geometry around it.
//
// A lane takes the table row of ITS line (spans into the packed line bytes + the entry slice) and emits the output
// text directly -- the Record is never materialised.  Two passes over the same emitter with different sinks:
//   k_encode<ENC, false>  count: sizes[i] = framed length of line i (0 when its decode or its encode failed; bit 31: no span
//                         holds a byte to escape) + status
//   k_block_scan + k_line_offsets   per-workgroup sums -> exclusive scan -> out_offsets[0..n]
//   k_encode<ENC, true>   write at out + out_offsets[i]
// so that the output is ONE contiguous, already framed byte stream in input order (what the outputs write).
// Input: the 64 lines of a workgroup are one contiguous byte range, staged into LDS with coalesced 16-byte loads; every
// HBM input of a lane (its offsets, its table row, its output slot) and the configuration mirror are requested BEFORE
// the tile so that one round trip covers them all (stage_tile_rider).  Spans without bytes that need escaping move
// sixteen bytes per LDS round trip (SWAR test per dword, fg_emit.hpp); the write sink assembles the byte stream into
// aligned SIXTEEN-byte blocks (emit::PackSink, round 5: a message's first and last block go out by dwords and bytes so that
// nothing outside the message is touched; rounds 1-4 packed dwords).  Output stores are per lane (each lane streams into its
// own message): per-lane 16-byte stores run at 3.0 TB/s against 0.92 TB/s for per-lane dword stores and 4.9 TB/s fully
// coalesced (tools/probe/store_patterns.cpp, profiles/r05c_store_patterns.log).  Measured (DESIGN.md sections 3.5, 4.1): the
// kernels run at two waves per SIMD (the LDS tile is the occupancy limit); a message goes out as ~45 pieces of up to sixteen
// bytes (round 5; 108 before), and what reaches HBM is 1.68x the message's bytes -- emitting once, wave-cooperatively, is what
// is left.
#include "fg_device.hpp"
#include "fg_emit.hpp"

// Build: the emitters are large force-inlined templates, so this file is compiled once per kernel with
// -DFG_ENC_TU=<FG_ENC_*> -DFG_ENC_TU_WRITE=<0|1> -DFG_ENC_TU_SLOTS=<n> (twelve objects, in parallel:
// flowgger_amd/build.py) and once without FG_ENC_TU for the scan kernels, the dispatcher and the C entry points.
namespace fg {

#define FG_ENC_ARGS                                                                                                             \
    const uint8_t *d_bytes, const uint64_t *d_offsets, uint64_t n, const DevTables &t, const EncCfg &cfg, uint32_t tile_cap,    \
        uint32_t cfg_lds, uint32_t *d_sizes, uint8_t *d_status, uint64_t *d_block_sums, const uint64_t *d_out_offsets,          \
        uint8_t *d_out, hipStream_t stream
// one definition per (encoder, pass, ranking slots) object
template <uint32_t ENC, bool WRITE, uint32_t SLOTS> int launch_encode_tu(FG_ENC_ARGS);

#ifdef FG_ENC_TU
// One 64-lane workgroup = 64 consecutive lines = ONE contiguous byte range of the packed buffer: it is staged into LDS
// with coalesced 16-byte loads (stage_tile) and every lane then reads ITS line out of LDS; a group whose bytes exceed
// the tile (long lines) reads from global memory instead (same emitter, GlobalReader).
// what a lane needs from HBM besides the tile: fetched before the tile is staged, every load in flight
struct LanePre {
    uint64_t o0;         // offsets[li]
    uint32_t meta;
    emit::RowRegs row;
    uint64_t oo0, oo1;   // WRITE: out_offsets[li], out_offsets[li + 1]
};
template <uint32_t ENC, bool WRITE, class R>
__device__ __forceinline__ void encode_lane(R rd, uint64_t li, const DevTables& t, const EncCfg& cfg, uint64_t* keys64, uint8_t* slot_ent,
                                            uint8_t* order, uint32_t* __restrict__ sizes, uint8_t* __restrict__ enc_status,
                                            const LanePre& pre, uint8_t* __restrict__ out, uint32_t* size_out) {
    if (WRITE) {
        emit::PackSink sink(out + pre.oo0);
        emit::row_write<ENC>(sink, pre.oo1 - pre.oo0, cfg, rd, t, li, pre.meta, keys64, slot_ent, order, &pre.row);
    } else {
        uint32_t st, plain;
        const uint32_t size = emit::row_size<ENC>(cfg, rd, t, li, pre.meta, keys64, slot_ent, order, &st, &pre.row, &plain);
        sizes[li] = size | plain << 31;  // (bit 31: no span of this row holds a byte to escape -- the write pass copies untested)
        if (enc_status) enc_status[li] = (uint8_t)st;
        *size_out = size;
    }
}

// A group whose bytes exceed the tile (long lines: rare) reads from global memory.  ITS OWN FUNCTION, NOT INLINED, everything BY VALUE
// (round 5): inlined, this instantiation of the emitters kept the emitter, the sink and -- through the references the emitter holds --
// the kernel's DevTables and EncCfg arguments in scratch memory for the WHOLE kernel (four allocas survive in the optimised IR;
// 1489 of the write kernel's 1939 scratch instructions belong to this path), and the hot path then fetched every table column
// pointer and configuration field from the stack copy: ~60 extra VMEM reads per wave, each of which waits behind the wave's
// outstanding output stores (vmcnt counts both) -- profiles/r05f_pmc_encode_*.json: the write kernel spent half its wave-cycles waiting.
template <uint32_t ENC, bool WRITE>
__device__ __attribute__((noinline)) uint32_t encode_lane_global(const uint8_t* bytes, uint64_t li, DevTables t, EncCfg cfg, uint64_t* keys64,
                                                                 uint8_t* slot_ent, uint8_t* order, uint32_t* sizes, uint8_t* enc_status,
                                                                 LanePre pre, uint8_t* out) {
    uint32_t size = 0;
    GlobalReader rd(reinterpret_cast<const uint32_t*>(bytes), pre.o0);
    encode_lane<ENC, WRITE>(rd, li, t, cfg, keys64, slot_ent, order, sizes, enc_status, pre, out, &size);
    return size;
}

// SLOTS = per-lane entries of the GELF key-ranking scratch (0 for the other encoders); cfg_lds = bytes of the
// configuration block [static keys | blob] to mirror in LDS (0: read it from global memory).
template <uint32_t ENC, bool WRITE, uint32_t SLOTS>
__global__ __launch_bounds__(kWave, 2) void k_encode(const uint8_t* __restrict__ bytes, const uint64_t* __restrict__ offsets, uint64_t n,
                                                 DevTables t, EncCfg cfg, uint32_t tile_cap, uint32_t cfg_lds, uint32_t* __restrict__ sizes,
                                                 uint8_t* __restrict__ enc_status, uint64_t* __restrict__ block_sums,
                                                 const uint64_t* __restrict__ out_offsets, uint8_t* __restrict__ out,
                                                 const uint4* __restrict__ cfg_block /* = cfg.keys, as a plain pointer */) {
    // (asynchronous form: the host never saw the total -- the capacity is checked here, by every workgroup, wave-uniform)
    if (WRITE && cfg.out_cap != 0ull && out_offsets[n] > cfg.out_cap) return;
    constexpr uint32_t kSlots = SLOTS ? SLOTS : 1u;
    __shared__ uint64_t s_keys[kWave * kSlots];
    __shared__ uint8_t s_slot[kWave * kSlots];
    __shared__ uint8_t s_order[kWave * kSlots];
    extern __shared__ __attribute__((aligned(16))) uint8_t s_tile[];  // tile_cap + 16 bytes, then the configuration mirror
    cfg.sort_slots = SLOTS;
    // every lane reads the same few hundred configuration bytes over and over: keep them in LDS (at most 4 KiB; its
    // loads ride along with the tile's, stage_tile_rider)
    uint8_t* s_cfg = s_tile + tile_cap + 16u;
    const uint64_t g0 = (uint64_t)blockIdx.x * kWave;
    const uint64_t g1 = g0 + kWave < n ? g0 + kWave : n;
    const uint64_t li = g0 + threadIdx.x;
    uint64_t* keys64 = s_keys + threadIdx.x * kSlots;
    uint8_t* slot_ent = s_slot + threadIdx.x * kSlots;
    uint8_t* order = s_order + threadIdx.x * kSlots;
    // Everything the lane needs from HBM first (offsets, table row, output slot), all loads in flight: the staging
    // below hides them.  (Fetched where they are used, each is a dependent round trip of the whole wave.)
    const bool live = li < n;
    const uint64_t lic = live ? li : n - 1u;
    LanePre pre;
    pre.o0 = offsets[lic];
    pre.meta = t.meta[lic];
    pre.row.load(t, lic);
    pre.oo0 = pre.oo1 = 0;
    if (WRITE) {
        pre.oo0 = out_offsets[lic];
        pre.oo1 = out_offsets[lic + 1u];
        pre.row.plain = sizes ? sizes[lic] >> 31 : 0u;  // (the count pass's verdict on this row's spans; no sizes: test again)
    }
    // the group's byte range (wave-uniform)
    const uint64_t a_begin = offsets[g0], a_end = offsets[g1];
    const uint64_t a0 = a_begin & ~15ull;
    const uint64_t span = (a_end - a0 + 15ull) & ~15ull;
    const bool staged = span <= (uint64_t)tile_cap;
    uint32_t size = 0;
    // (20 KiB per round trip: the usual tile in one.  A group that exceeds the tile reads from global memory: only the
    //  mirror is staged -- without a mirror, cfg_lds == 0, one chunk lands in the 16 spare bytes the launch reserves.)
    stage_tile_rider<20>(bytes, a0, staged ? (uint32_t)span : 0u, s_tile, cfg_block, cfg_lds >> 4, reinterpret_cast<uint4*>(s_cfg));
    {
        const uint8_t* g_cfg = reinterpret_cast<const uint8_t*>(cfg.keys);
        cfg.blob = cfg_lds ? s_cfg + (cfg.blob - g_cfg) : cfg.blob;
        cfg.keys = cfg_lds ? reinterpret_cast<const StaticKey*>(s_cfg) : cfg.keys;
    }
    __syncthreads();  // single-wave workgroup: orders the LDS writes before the lanes' reads
    if (staged) {
        if (live) {
            LdsReader rd(reinterpret_cast<const uint32_t*>(s_tile), (uint32_t)(pre.o0 - a0));
            encode_lane<ENC, WRITE>(rd, li, t, cfg, keys64, slot_ent, order, sizes, enc_status, pre, out, &size);
        }
    } else if (live) {
        size = encode_lane_global<ENC, WRITE>(bytes, li, t, cfg, keys64, slot_ent, order, sizes, enc_status, pre, out);
    }
    if (!WRITE) {  // the 64 lines of this workgroup: one partial sum for the offset scan
        uint64_t sum = size;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d, kWave);
        if (threadIdx.x == 0) block_sums[blockIdx.x] = sum;
    }
}

template <>
int launch_encode_tu<FG_ENC_TU, (FG_ENC_TU_WRITE != 0), FG_ENC_TU_SLOTS>(FG_ENC_ARGS) {
    const uint64_t blocks = (n + kWave - 1) / kWave;
    if (blocks > 0x7FFFFFFFull) return -1;
    hipLaunchKernelGGL((k_encode<FG_ENC_TU, (FG_ENC_TU_WRITE != 0), FG_ENC_TU_SLOTS>), dim3((uint32_t)blocks), dim3(kWave),
                       tile_cap + 16u + (cfg_lds ? cfg_lds : 16u), stream, d_bytes, d_offsets, n, t, cfg, tile_cap, cfg_lds, d_sizes, d_status, d_block_sums,
                       d_out_offsets, d_out, reinterpret_cast<const uint4*>(cfg.keys));
    return 0;
}
#else  // !FG_ENC_TU: scan kernels, dispatcher, C entry points

// exclusive scan of the per-workgroup sums (nb = ceil(n / 64) of them) in place; off_n[0] = the grand total.  One
// workgroup; every thread owns kScanPerThread consecutive sums per round (8192 sums per round: 100 M lines = 191 rounds).
constexpr uint32_t kScanPerThread = 8;
// `base` = where this batch's (this slice's) first message goes: offsets come out absolute.
__global__ __launch_bounds__(1024) void k_block_scan(uint64_t* __restrict__ block_sums, uint64_t nb, uint64_t* __restrict__ off_n,
                                                     uint64_t base) {
    __shared__ uint64_t wave_tot[16];
    __shared__ uint64_t carry_s;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    if (tid == 0) carry_s = base;
    __syncthreads();
    const uint64_t last = nb ? nb - 1u : 0u;
    for (uint64_t b0 = 0; b0 < nb; b0 += 1024ull * kScanPerThread) {
        const uint64_t i0 = b0 + (uint64_t)tid * kScanPerThread;
        uint64_t x[kScanPerThread];
        uint64_t t = 0;
#pragma unroll
        for (uint32_t v = 0; v < kScanPerThread; ++v) {  // (unconditional, index-clamped loads: all in flight)
            const uint64_t i = i0 + v;
            const uint64_t val = block_sums[i < last ? i : last];
            x[v] = i < nb ? val : 0ull;
            t += x[v];
        }
        uint64_t inc = t;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            uint64_t y = __shfl_up(inc, d, 64);
            if (lane >= (uint32_t)d) inc += y;
        }
        if (lane == 63) wave_tot[wv] = inc;
        __syncthreads();
        uint64_t o = carry_s;
        for (uint32_t w = 0; w < wv; ++w) o += wave_tot[w];
        uint64_t run = o + inc - t;  // exclusive prefix of this thread's first sum
#pragma unroll
        for (uint32_t v = 0; v < kScanPerThread; ++v) {
            const uint64_t i = i0 + v;
            if (i < nb) block_sums[i] = run;
            run += x[v];
        }
        __syncthreads();
        if (tid == 1023) carry_s = o + inc;
        __syncthreads();
    }
    if (tid == 0) *off_n = carry_s;
}
// off[i] = block offset + exclusive scan of the sizes inside the 64-line block
__global__ __launch_bounds__(kWave) void k_line_offsets(const uint32_t* __restrict__ sizes, const uint64_t* __restrict__ block_off, uint64_t n,
                                                       uint64_t* __restrict__ off) {
    const uint64_t li = (uint64_t)blockIdx.x * kWave + threadIdx.x;
    const uint32_t x = li < n ? sizes[li] & 0x7FFFFFFFu : 0u;  // (bit 31 is the count pass's note for the write pass: k_encode)
    uint32_t tot;
    const uint32_t ex = wave_exclusive_sum(x, &tot);
    if (li < n) off[li] = block_off[blockIdx.x] + ex;
}

template <bool WRITE>
static int launch_encode(FG_ENC_ARGS) {
#define FG_CALL(E, SL) \
    return launch_encode_tu<E, WRITE, SL>(d_bytes, d_offsets, n, t, cfg, tile_cap, cfg_lds, d_sizes, d_status, d_block_sums, d_out_offsets, d_out, stream)
    switch (cfg.enc) {
        case FG_ENC_GELF:
            // few pairs per line on average: a small ranking scratch (more LDS left for occupancy); a line with more
            // pairs than slots takes the exact selection path
            static_assert(emit::kSortSlots == 32u, "flowgger_amd/build.py compiles the GELF kernels for 1, 8 and 32 slots");
            if (cfg.sort_slots <= 1u) FG_CALL(FG_ENC_GELF, 1u);  // no pairs in the batch: 4.5 KiB of LDS back for occupancy
            if (cfg.sort_slots <= 8u) FG_CALL(FG_ENC_GELF, 8u);
            FG_CALL(FG_ENC_GELF, 32u);
        case FG_ENC_LTSV: FG_CALL(FG_ENC_LTSV, 0u);
        case FG_ENC_RFC5424: FG_CALL(FG_ENC_RFC5424, 0u);
        case FG_ENC_RFC3164: FG_CALL(FG_ENC_RFC3164, 0u);
        case FG_ENC_PASSTHROUGH: FG_CALL(FG_ENC_PASSTHROUGH, 0u);
        default: return -1;
    }
#undef FG_CALL
}
#endif  // FG_ENC_TU

}  // namespace fg

#ifndef FG_ENC_TU
// d_sizes: n u32; d_block_sums: ceil(n / 64) u64 (scratch).  Two steps so that a caller that encodes a batch slice by slice can
// queue the (long) count kernel of a slice before it knows where the slice's output starts:
//   fg_launch_encode_count   sizes per line + per-64-line sums
//   fg_launch_encode_scan    sums -> out_offsets[0 .. n], absolute from `base`; out_offsets[n] = base + the slice's bytes
extern "C" int fg_launch_encode_count(const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t n, const fg::DevTables* t,
                                      const fg::EncCfg* cfg, uint32_t tile_cap, uint32_t cfg_lds, uint32_t* d_sizes,
                                      uint64_t* d_block_sums, uint8_t* d_status, hipStream_t stream) {
    if (n == 0) return 0;
    if (fg::launch_encode<false>(d_bytes, d_offsets, n, *t, *cfg, tile_cap, cfg_lds, d_sizes, d_status, d_block_sums, nullptr, nullptr, stream) != 0) return -1;
    return (int)hipGetLastError();
}
extern "C" int fg_launch_encode_scan(const uint32_t* d_sizes, uint64_t* d_block_sums, uint64_t n, uint64_t* d_out_offsets, uint64_t base,
                                     hipStream_t stream) {
    if (n == 0) return 0;
    const uint64_t nb = (n + fg::kWave - 1) / fg::kWave;
    hipLaunchKernelGGL(fg::k_block_scan, dim3(1), dim3(1024), 0, stream, d_block_sums, nb, d_out_offsets + n, base);
    hipLaunchKernelGGL(fg::k_line_offsets, dim3((uint32_t)nb), dim3(fg::kWave), 0, stream, d_sizes, d_block_sums, n, d_out_offsets);
    return (int)hipGetLastError();
}
extern "C" int fg_launch_encode_sizes(const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t n, const fg::DevTables* t,
                                      const fg::EncCfg* cfg, uint32_t tile_cap, uint32_t cfg_lds, uint32_t* d_sizes,
                                      uint64_t* d_block_sums, uint8_t* d_status, uint64_t* d_out_offsets, hipStream_t stream) {
    if (n == 0) return 0;
    if (fg_launch_encode_count(d_bytes, d_offsets, n, t, cfg, tile_cap, cfg_lds, d_sizes, d_block_sums, d_status, stream) != 0) return -1;
    return fg_launch_encode_scan(d_sizes, d_block_sums, n, d_out_offsets, 0ull, stream);
}
extern "C" int fg_launch_encode_write(const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t n, const fg::DevTables* t,
                                      const fg::EncCfg* cfg, uint32_t tile_cap, uint32_t cfg_lds, const uint64_t* d_out_offsets,
                                      uint8_t* d_out, const uint32_t* d_sizes, hipStream_t stream) {
    if (n == 0) return 0;
    // (d_sizes: what fg_launch_encode_count left for these n rows -- read for its bit 31 only; null = every span is tested again)
    if (fg::launch_encode<true>(d_bytes, d_offsets, n, *t, *cfg, tile_cap, cfg_lds, const_cast<uint32_t*>(d_sizes), nullptr, nullptr, d_out_offsets, d_out, stream) != 0) return -1;
    return (int)hipGetLastError();
}
#endif  // !FG_ENC_TU
