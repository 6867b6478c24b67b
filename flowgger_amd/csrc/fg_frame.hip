// fg_frame.hip -- gfx950 kernels for the step BEFORE Decoder::decode: framing a raw byte stream
// into lines and rejecting lines that are not valid UTF-8.
//
// Reference semantics:
//   LineSplitter  `for line in buf_reader.lines()`   src/flowgger/splitter/line_splitter.rs:17-25
//       BufRead::lines(): split on '\n'; the '\n' and ONE preceding '\r' are not part of the line;
//       a final unterminated piece is a line if it is non-empty; invalid UTF-8 -> InvalidData ->
//       stderr "Invalid UTF-8 input", the line is dropped.
//   NulSplitter   `for line in buf_reader.split(0)` + str::from_utf8   nul_splitter.rs:18-40
//   (SyslenSplitter's "<len> " prefix chain is sequential per connection and stays on the host.)
//
// Output: frame i = bytes[offsets[i] .. offsets[i+1]) INCLUDING its terminator (the decode
// kernels strip "\n" / "\r\n" / "\0" themselves, fg_decode_frames_device), bad_utf8[i] = 1 when
// the frame is not valid UTF-8.
//
// Two streaming passes, HBM-bound:
//   scan  every byte once: 16 B per lane, per 16-byte chunk a delimiter mask and a UTF-8 error
//         mask (both 16 bits, one u32 store), delimiter count per 16 KiB block;
//   (a one-workgroup exclusive scan of the block counts;)
//   emit  reads only the masks (1/4 of the input): rank of every delimiter -> offsets[], error
//         bits -> bad_utf8[frame].
// UTF-8 validity is judged per byte position from the byte and its three predecessors (the
// table-free form of the well-formedness rules, Unicode 15 Table 3-7), so blocks, waves and
// lanes need no carried state; an error is always flagged inside the frame it belongs to
// because the terminators are ASCII.
#include "fg_pipeline.hpp"

namespace fg {

constexpr uint32_t kFrameBlock = 16384;  // bytes per wave-block
constexpr uint32_t kFrameRows = kFrameBlock / (16 * kWave);  // 16

// bit7 flags, per byte of a dword
__device__ __forceinline__ uint32_t f_cont(uint32_t b) { return b & ~(b << 1) & 0x80808080u; }          // 80..BF
__device__ __forceinline__ uint32_t f_ge_c0(uint32_t b) { return b & (b << 1) & 0x80808080u; }          // C0..FF
__device__ __forceinline__ uint32_t f_ge_e0(uint32_t b) { return b & (b << 1) & (b << 2) & 0x80808080u; }
__device__ __forceinline__ uint32_t f_ge_f0(uint32_t b) { return b & (b << 1) & (b << 2) & (b << 3) & 0x80808080u; }
__device__ __forceinline__ uint32_t f_eq_hi(uint32_t b, uint32_t pat) {  // byte == pat's byte (pat bytes >= 0x80)
    uint32_t y = b ^ pat;                                                // zero byte <=> equal
    return ~(((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y) & 0x80808080u;
}
// UTF-8 errors of the four bytes in `b`, given the previous dword `p` (bytes just before b):
// bit7 of byte i set <=> the sequence rules are violated AT byte i.
__device__ __forceinline__ uint32_t utf8_err_flags(uint32_t b, uint32_t p) {
    const uint32_t p1 = __builtin_amdgcn_alignbyte(b, p, 3);  // byte i-1 for every i
    const uint32_t p2 = __builtin_amdgcn_alignbyte(b, p, 2);
    const uint32_t p3 = __builtin_amdgcn_alignbyte(b, p, 1);
    // a continuation byte is due here: after a lead, or as the 3rd / 4th byte of a sequence whose
    // earlier bytes WERE continuations (a sequence broken earlier was flagged there and expects
    // nothing more -- so no expectation ever crosses an ASCII byte such as the frame terminator)
    const uint32_t c1 = f_cont(p1), c2 = f_cont(p2);
    const uint32_t must = f_ge_c0(p1) | (f_ge_e0(p2) & c1) | (f_ge_f0(p3) & c2 & c1);
    uint32_t err = (f_cont(b) ^ must);                                   // missing or stray continuation
    // bytes that never appear: C0, C1, F5..FF
    const uint32_t y = (b & 0xFEFEFEFEu) ^ 0xC0C0C0C0u;
    err |= ~(((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y) & 0x80808080u;
    err |= b & ((b & 0x7F7F7F7Fu) + 0x0B0B0B0Bu) & 0x80808080u;          // >= F5
    // second-byte ranges: E0 A0..BF | ED 80..9F | F0 90..BF | F4 80..8F
    const uint32_t b5 = b << 2, b4 = b << 3;                             // bit5 / bit4 of each byte moved to bit7
    err |= f_eq_hi(p1, 0xE0E0E0E0u) & ~b5;
    err |= f_eq_hi(p1, 0xEDEDEDEDu) & b5;
    err |= f_eq_hi(p1, 0xF0F0F0F0u) & ~b5 & ~b4;
    err |= f_eq_hi(p1, 0xF4F4F4F4u) & (b5 | b4);
    return err & 0x80808080u;
}

// pass 1.  masks[chunk] = delimiter mask | error mask << 16 (chunk = 16 bytes); counts[block].
// (blk0: first block of the range this launch covers -- a SLICE of the stream whose bytes have arrived; the pipelined host path frames
//  slice by slice while the next one is still on the link)
// COPY: the stream is read where the caller has it -- pinned host memory, over the link -- and written to its place in HBM on the
// way (copy_to + the same offsets): the upload of the raw-stream host path without the copy engine (fg_host_pipeline.cpp).
template <bool COPY>
__global__ __launch_bounds__(kWave) void k_frame_scan(const uint8_t* __restrict__ bytes, uint64_t nbytes, uint32_t delim_pat,
                                                     uint32_t* __restrict__ masks, uint32_t* __restrict__ counts, uint64_t blk0,
                                                     uint8_t* __restrict__ copy_to) {
    const uint32_t lane = threadIdx.x;
    const uint64_t blk = blk0 + blockIdx.x;
    const uint64_t base = blk * kFrameBlock;
    const uint64_t left = nbytes - base;  // the last block may be empty: it only carries the "cut off by the end" check
    const uint32_t span = left >= kFrameBlock ? kFrameBlock : (uint32_t)((left + 15u) & ~15ull);
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(bytes + base), (short)0, (int)span, 0x00020000);
    u32x4 v[kFrameRows];
#pragma unroll
    for (int k = 0; k < (int)kFrameRows; ++k)
        v[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(lane * 16u + k * 1024u), 0, COPY ? FG_STREAM_AUX : 0);  // (COPY: read once, over the link)
    if constexpr (COPY) {
        __amdgpu_buffer_rsrc_t dst = __builtin_amdgcn_make_buffer_rsrc(copy_to + base, (short)0, (int)span, 0x00020000);
#pragma unroll
        for (int k = 0; k < (int)kFrameRows; ++k) __builtin_amdgcn_raw_buffer_store_b128(v[k], dst, (int)(lane * 16u + k * 1024u), 0, 0);
    }
    // the dword just before this block (row 0, lane 0 needs it)
    uint32_t before = 0;
    if (base != 0) before = *reinterpret_cast<const uint32_t*>(bytes + base - 4);
    uint32_t total = 0;
    uint32_t prev_row_last = before;
#pragma unroll
    for (int k = 0; k < (int)kFrameRows; ++k) {
        const uint4 q = make_uint4(v[k][0], v[k][1], v[k][2], v[k][3]);
        // bytes of this chunk inside the stream: valid positions 0..nv-1; position nv (one past the
        // end of the stream) may carry the "sequence cut off by the end" error bit
        const uint64_t cpos = base + (uint64_t)(k * kWave + lane) * 16u;
        const uint32_t nv = cpos >= nbytes ? 0u : (nbytes - cpos >= 16u ? 16u : (uint32_t)(nbytes - cpos));
        uint32_t dm = mask16_eq(q, delim_pat) & ((1u << nv) - 1u);
        // previous dword: lane-1's last dword; lane 0: last dword of the previous row / block
        uint32_t pw = __shfl_up(q.w, 1, kWave);
        const uint32_t row_last = __shfl(q.w, kWave - 1, kWave);
        if (lane == 0) pw = prev_row_last;
        prev_row_last = row_last;
        uint32_t em = 0;
        if (((q.x | q.y | q.z | q.w | pw) & 0x80808080u) != 0u) {
            em = gather16(utf8_err_flags(q.x, pw), utf8_err_flags(q.y, q.x), utf8_err_flags(q.z, q.y), utf8_err_flags(q.w, q.z));
            em &= nv >= 16u ? 0xFFFFu : ((2u << nv) - 1u);  // keep position nv itself (zero fill = "not a continuation")
            if (cpos > nbytes) em = 0;
        }
        masks[blk * (kFrameBlock / 16u) + k * kWave + lane] = dm | (em << 16);
        total += (uint32_t)__builtin_popcount(dm);
    }
    // wave sum -> counts[blk]
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) total += __shfl_xor(total, d, kWave);
    if (lane == 0) counts[blk] = total;
}

// exclusive scan of counts[0..nblk) -> pref[0..nblk], pref[nblk] = total; one workgroup.  carry_in: the scan continues the one
// of the blocks before (pref[0] already holds their total: the previous slice's pref[nblk]).
__global__ __launch_bounds__(1024) void k_frame_prefix(const uint32_t* __restrict__ counts, uint64_t nblk, uint64_t* __restrict__ pref,
                                                       uint32_t carry_in) {
    __shared__ uint64_t wave_tot[16];
    __shared__ uint64_t carry_s;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    if (tid == 0) carry_s = carry_in ? pref[0] : 0;
    __syncthreads();
    for (uint64_t b0 = 0; b0 < nblk; b0 += 1024) {
        const uint64_t i = b0 + tid;
        uint64_t x = i < nblk ? counts[i] : 0;
        uint64_t inc = x;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            uint64_t y = __shfl_up(inc, d, 64);
            if (lane >= (uint32_t)d) inc += y;
        }
        if (lane == 63) wave_tot[wv] = inc;
        __syncthreads();
        uint64_t off = carry_s;
        for (uint32_t w = 0; w < wv; ++w) off += wave_tot[w];
        if (i < nblk) pref[i] = off + inc - x;
        __syncthreads();
        if (tid == 1023) carry_s = off + inc;
        __syncthreads();
    }
    if (tid == 0) pref[nblk] = carry_s;
}

// pass 2.  A lane owns 16 consecutive chunks (256 bytes) of its block.
// (blk0 / whole: see k_frame_scan; a slice launch leaves the end of a final unterminated frame to the host, which knows the total
//  only after the last slice)
__global__ __launch_bounds__(kWave) void k_frame_emit(const uint32_t* __restrict__ masks, const uint64_t* __restrict__ pref,
                                                     uint64_t nbytes, uint64_t nblk, uint64_t* __restrict__ offsets,
                                                     uint8_t* __restrict__ bad, uint64_t cap, uint64_t blk0, uint32_t whole) {
    const uint32_t lane = threadIdx.x;
    const uint64_t blk = blk0 + blockIdx.x;
    const uint4* src = reinterpret_cast<const uint4*>(masks + blk * (kFrameBlock / 16u) + lane * 16u);
    uint4 m[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) m[k] = src[k];
    uint32_t w[16] = {m[0].x, m[0].y, m[0].z, m[0].w, m[1].x, m[1].y, m[1].z, m[1].w,
                      m[2].x, m[2].y, m[2].z, m[2].w, m[3].x, m[3].y, m[3].z, m[3].w};
    uint32_t mine = 0, any_err = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        mine += (uint32_t)__builtin_popcount(w[j] & 0xFFFFu);
        any_err |= w[j] >> 16;
    }
    uint32_t total;
    const uint32_t ex = wave_exclusive_sum(mine, &total);
    uint64_t rank = pref[blk] + ex;  // delimiters before this lane's first byte
    if (blk == 0 && lane == 0) {
        offsets[0] = 0;
        if (whole) {
            const uint64_t total_delims = pref[nblk];
            if (total_delims + 1 <= cap) offsets[total_delims + 1] = nbytes;  // end of a final unterminated frame
        }
    }
    if (mine == 0 && any_err == 0) return;
    const uint64_t lane_base = blk * (uint64_t)kFrameBlock + (uint64_t)lane * 256u;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        uint32_t dm = w[j] & 0xFFFFu, em = w[j] >> 16;
        const uint64_t cpos = lane_base + (uint64_t)j * 16u;
        while (em) {  // rare
            const uint32_t b = (uint32_t)__builtin_ctz(em);
            em &= em - 1u;
            const uint64_t frame = rank + (uint32_t)__builtin_popcount(dm & ((1u << b) - 1u));
            if (frame < cap) bad[frame] = 1;
        }
        while (dm) {
            const uint32_t b = (uint32_t)__builtin_ctz(dm);
            dm &= dm - 1u;
            ++rank;
            if (rank <= cap) offsets[rank] = cpos + b + 1u;
        }
    }
}

}  // namespace fg

// scratch: masks = ceil(nbytes / 16 KiB) * 1024 u32, counts = nblk u32, pref = (nblk + 1) u64
// blocks: one more than needed for the bytes when nbytes is a multiple of the block size, so
// that position `nbytes` itself (where a sequence cut off by the end of the stream is flagged) is
// always covered
static inline uint64_t frame_blocks(uint64_t nbytes) { return nbytes / fg::kFrameBlock + 1; }

extern "C" uint64_t fg_frame_scratch_bytes(uint64_t nbytes) {
    const uint64_t nblk = frame_blocks(nbytes);
    return nblk * 4096u + ((nblk * 4u + 255u) & ~255ull) + (nblk + 1u) * 8u + 256u;
}
extern "C" int fg_launch_frame(const uint8_t* d_bytes, uint64_t nbytes, uint32_t delim, uint8_t* scratch, uint64_t* d_offsets,
                               uint8_t* d_bad, uint64_t cap, uint64_t** d_total_out, hipStream_t stream) {
    const uint64_t nblk = frame_blocks(nbytes);
    if (nblk > 0x7FFFFFFFull) return -1;
    uint32_t* masks = reinterpret_cast<uint32_t*>(scratch);
    uint32_t* counts = reinterpret_cast<uint32_t*>(scratch + nblk * 4096u);
    uint64_t* pref = reinterpret_cast<uint64_t*>(scratch + nblk * 4096u + ((nblk * 4u + 255u) & ~255ull));
    const uint32_t pat = delim * 0x01010101u;
    (void)hipMemsetAsync(d_bad, 0, cap, stream);
    hipLaunchKernelGGL(fg::k_frame_scan<false>, dim3((uint32_t)nblk), dim3(fg::kWave), 0, stream, d_bytes, nbytes, pat, masks, counts, (uint64_t)0,
                       (uint8_t*)nullptr);
    hipLaunchKernelGGL(fg::k_frame_prefix, dim3(1), dim3(1024), 0, stream, counts, nblk, pref, 0u);
    hipLaunchKernelGGL(fg::k_frame_emit, dim3((uint32_t)nblk), dim3(fg::kWave), 0, stream, masks, pref, nbytes, nblk, d_offsets,
                       d_bad, cap, (uint64_t)0, 1u);
    *d_total_out = pref + nblk;
    return (int)hipGetLastError();
}

// One SLICE of the stream: the blocks [blk0, blk1) (slice boundaries are multiples of the 16 KiB block; the last slice ends at
// frame_blocks(nbytes)), whose bytes -- and everything before them -- are in d_bytes.  Continues the delimiter ranks where the
// slice before stopped; *d_total_out = the device word that holds the delimiters up to the end of this slice.  d_bad must have
// been cleared for the whole batch beforehand; offsets[total + 1] of a final unterminated frame is the caller's business.
extern "C" uint64_t fg_frame_block_bytes(void) { return fg::kFrameBlock; }
// src != null: the slice's bytes are read from `src` (the device view of the caller's pinned buffer, same offsets) and stored to d_bytes
// by the scan itself -- no upload has to have happened.
extern "C" int fg_launch_frame_slice(const uint8_t* d_bytes, uint64_t nbytes, uint32_t delim, uint8_t* scratch, uint64_t* d_offsets,
                                     uint8_t* d_bad, uint64_t cap, uint64_t blk0, uint64_t blk1, uint64_t** d_total_out,
                                     hipStream_t stream, const uint8_t* src) {
    const uint64_t nblk = frame_blocks(nbytes);
    if (nblk > 0x7FFFFFFFull || blk1 > nblk || blk0 >= blk1) return -1;
    uint32_t* masks = reinterpret_cast<uint32_t*>(scratch);
    uint32_t* counts = reinterpret_cast<uint32_t*>(scratch + nblk * 4096u);
    uint64_t* pref = reinterpret_cast<uint64_t*>(scratch + nblk * 4096u + ((nblk * 4u + 255u) & ~255ull));
    const uint32_t pat = delim * 0x01010101u;
    const uint32_t nb = (uint32_t)(blk1 - blk0);
    if (src)
        hipLaunchKernelGGL(fg::k_frame_scan<true>, dim3(nb), dim3(fg::kWave), 0, stream, src, nbytes, pat, masks, counts, blk0,
                           const_cast<uint8_t*>(d_bytes));
    else
        hipLaunchKernelGGL(fg::k_frame_scan<false>, dim3(nb), dim3(fg::kWave), 0, stream, d_bytes, nbytes, pat, masks, counts, blk0, (uint8_t*)nullptr);
    hipLaunchKernelGGL(fg::k_frame_prefix, dim3(1), dim3(1024), 0, stream, counts + blk0, (uint64_t)nb, pref + blk0, blk0 ? 1u : 0u);
    hipLaunchKernelGGL(fg::k_frame_emit, dim3(nb), dim3(fg::kWave), 0, stream, masks, pref, nbytes, nblk, d_offsets, d_bad, cap, blk0, 0u);
    *d_total_out = pref + blk1;
    return (int)hipGetLastError();
}

// One 8-byte word from device memory into ANY device-addressable memory (a pinned host word included) by a kernel: the sliced host
// paths bring their per-slice counters back this way, because a hipMemcpy of 8 bytes queues on the same copy engine as the 32 MiB
// uploads that were issued ahead of it and would wait for all of them (profiles/r04a_timeline_*).
namespace fg {
__global__ void k_poke64(const uint64_t* __restrict__ src, uint64_t* __restrict__ dst) {
    if (threadIdx.x == 0) {
        const uint64_t v = *src;
        __atomic_store_n(dst, v, __ATOMIC_RELAXED);
        __threadfence_system();
    }
}
}  // namespace fg
extern "C" int fg_launch_poke64(const uint64_t* d_src, uint64_t* dst_devview, hipStream_t stream) {
    hipLaunchKernelGGL(fg::k_poke64, dim3(1), dim3(64), 0, stream, d_src, dst_devview);
    return (int)hipGetLastError();
}

