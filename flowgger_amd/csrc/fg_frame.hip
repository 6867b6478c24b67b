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
// ONE streaming pass (k_frame_onepass, round 4): a wave takes a 16 KiB tile -- 16 B per lane and row, per 16-byte chunk a delimiter
// mask and a UTF-8 error mask -- counts its delimiters, learns how many lie before its tile from the tiles before it (a chained
// scan with decoupled look-back over one 8-byte descriptor per tile: flag | count), and writes the offsets of ITS delimiters and the
// bad-UTF-8 flags of ITS error positions from the masks it still holds (transposed through 4 KiB of LDS so that a lane owns 256
// consecutive bytes).  The stream is read once; nothing but the descriptors is written besides the output.
// The classic form (k_frame_scan -> k_frame_prefix -> k_frame_emit: masks to HBM, a one-workgroup scan of the tile counts, a second
// pass over the masks; 39 GB of traffic for a 25.5 GB stream) is kept as the fall-back: a wave that waits on a predecessor's
// descriptor longer than kSpinLimit polls raises an abort word and leaves the sentinel as the range's total, the launcher's caller
// sees it and runs the classic kernels (forward progress of the chain rests on workgroups being dispatched in index order; the bound makes a
// platform where that ever fails slow, not hung).
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


// ---- the one-pass form ----------------------------------------------------------------------------------------------------
// desc[tile] = flag << 62 | value: flag 0 nothing yet, 1 the tile's own delimiter count (A), 2 the count of all tiles up to and
// including this one (P).  Zeroed once per stream (fg_launch_frame / the first slice).
constexpr uint64_t kDescA = 1ull << 62, kDescP = 2ull << 62, kDescVal = (1ull << 62) - 1ull;
// polls before a waiting tile gives up: a poll is an s_sleep(2) plus an agent-scope load that goes to L2, ~0.5-1 us, so 2^18 polls are
// a fifth of a second -- against a look-back that normally resolves within a few polls.  (2^22 were SECONDS per waiting tile)
constexpr uint32_t kSpinLimit = 1u << 18;
constexpr uint32_t kWholeStream = 1u, kSelfTestStall = 2u;  // bits of the kernel's `whole` argument
constexpr uint64_t kFrameAborted = ~0ull;  // what pref[blk1] holds when the chain gave up (the caller runs the classic kernels)

// A workgroup of kTileWaves = eight waves takes a 128 KiB tile (a wave = one 16 KiB block of it, as in the classic scan); ONE descriptor per tile:
// the chain moves at most 64 descriptors per look-back step (a trip to L2), so the tile must be large enough for that to outrun
// the memory system -- with one wave per 16 KiB descriptor the scan ran at 2.9 TB/s, bound by exactly that (profiles/r04x_*).
constexpr uint32_t kTileWaves = 8;
template <bool COPY>
__global__ __launch_bounds__(kWave* kTileWaves) void k_frame_onepass(const uint8_t* __restrict__ bytes, uint64_t nbytes, uint32_t delim_pat,
                                                                    uint64_t* __restrict__ desc, uint64_t* __restrict__ pref,
                                                                    uint32_t* __restrict__ abort_w, uint64_t* __restrict__ offsets,
                                                                    uint8_t* __restrict__ bad, uint64_t cap, uint64_t tile0, uint64_t blk1,
                                                                    uint64_t nblk, uint32_t whole, uint8_t* __restrict__ copy_to) {
    __shared__ __attribute__((aligned(16))) uint32_t m_lds[kTileWaves][kFrameBlock / 16u];
    __shared__ uint32_t s_cnt[kTileWaves];
    __shared__ uint64_t s_prefix;
    __shared__ uint32_t s_abort;
    const uint32_t lane = threadIdx.x & (kWave - 1u), wv = threadIdx.x / kWave;
    const uint64_t tile = tile0 + blockIdx.x;
    const uint64_t blk = tile * kTileWaves + wv;  // this wave's 16 KiB block (may lie behind the stream: empty then)
    const uint64_t base = blk * kFrameBlock;
    const uint64_t left = base < nbytes ? nbytes - base : 0ull;  // (a block AT the end only carries the "cut off by the end" check)
    const uint32_t span = left >= kFrameBlock ? kFrameBlock : (uint32_t)((left + 15u) & ~15ull);
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(bytes + (base < nbytes ? base : 0ull)), (short)0, (int)span, 0x00020000);
    u32x4 v[kFrameRows];
#pragma unroll
    for (int k = 0; k < (int)kFrameRows; ++k)
        v[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(lane * 16u + k * 1024u), 0, COPY ? FG_STREAM_AUX : 0);
    if constexpr (COPY) {
        __amdgpu_buffer_rsrc_t dst = __builtin_amdgcn_make_buffer_rsrc(copy_to + (base < nbytes ? base : 0ull), (short)0, (int)span, 0x00020000);
#pragma unroll
        for (int k = 0; k < (int)kFrameRows; ++k) __builtin_amdgcn_raw_buffer_store_b128(v[k], dst, (int)(lane * 16u + k * 1024u), 0, 0);
    }
    uint32_t before = 0;  // the dword just before this block (row 0, lane 0 needs it)
    if (base != 0 && base <= nbytes) before = *reinterpret_cast<const uint32_t*>(bytes + base - 4);
    uint32_t count = 0;
    uint32_t prev_row_last = before;
#pragma unroll
    for (int k = 0; k < (int)kFrameRows; ++k) {
        const uint4 q = make_uint4(v[k][0], v[k][1], v[k][2], v[k][3]);
        const uint64_t cpos = base + (uint64_t)(k * kWave + lane) * 16u;
        const uint32_t nv = cpos >= nbytes ? 0u : (nbytes - cpos >= 16u ? 16u : (uint32_t)(nbytes - cpos));
        const uint32_t dm = mask16_eq(q, delim_pat) & ((1u << nv) - 1u);
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
        m_lds[wv][k * kWave + lane] = dm | (em << 16);
        count += (uint32_t)__builtin_popcount(dm);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) count += __shfl_xor(count, d, kWave);
    if (lane == 0) s_cnt[wv] = count;
    if (threadIdx.x == 0) s_abort = 0u;
    __syncthreads();
    uint32_t tile_count = 0, wave_off = 0;
#pragma unroll
    for (uint32_t k = 0; k < kTileWaves; ++k) {
        const uint32_t c = s_cnt[k];
        wave_off += k < wv ? c : 0u;
        tile_count += c;
    }
    // the tile's own count, for the tiles behind it (the first tile of the stream knows its prefix)
    // (self-test, FG_LO_FRAME_SELFTEST_STALL: the second tile of the launch never publishes anything -- every tile behind it must give up
    //  within the spin bound and the caller must fall back to the classic kernels)
    const bool stall = (whole & kSelfTestStall) != 0u && blockIdx.x == 1u;
    if (threadIdx.x == 0 && !stall) __hip_atomic_store(desc + tile, (tile == 0 ? kDescP : kDescA) | (uint64_t)tile_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // a lane now owns 16 consecutive chunks (256 bytes) of its wave's block: rank of its first delimiter inside the tile
    uint32_t w[16];
    {
        const uint4* src = reinterpret_cast<const uint4*>(m_lds[wv] + lane * 16u);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint4 m = src[k];
            w[4 * k] = m.x; w[4 * k + 1] = m.y; w[4 * k + 2] = m.z; w[4 * k + 3] = m.w;
        }
    }
    uint32_t mine = 0, any_err = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        mine += (uint32_t)__builtin_popcount(w[j] & 0xFFFFu);
        any_err |= w[j] >> 16;
    }
    uint32_t tot;
    const uint32_t ex = wave_exclusive_sum(mine, &tot);
    // ---- look-back (the tile's first wave): the delimiters of all tiles before this one ----
    if (wv == 0) {
        uint64_t prefix = 0;
        bool aborted = false;
        if (tile != 0) {
            int64_t idx = (int64_t)tile - 1;
            uint32_t polls = 0;
            for (;;) {
                const int64_t j = idx - (int64_t)lane;  // lane i looks at tile idx - i; before the stream: a prefix of zero
                uint64_t d = kDescP;
                if (j >= 0) d = __hip_atomic_load(desc + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t fl = (uint32_t)(d >> 62);
                const uint64_t mp = __ballot(fl == 2u), mx = __ballot(fl == 0u);
                const uint32_t p = mp ? (uint32_t)__builtin_ctzll(mp) : 64u;  // the nearest tile that knows its prefix
                const uint64_t upto = p >= 63u ? ~0ull : ((2ull << p) - 1ull);
                if (mx & upto) {  // a tile between here and there has not even counted yet: wait for it
                    ++polls;
                    if (polls > kSpinLimit) {
                        if (lane == 0) __hip_atomic_store(abort_w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        aborted = true;
                    } else if ((polls & 15u) == 0u && __hip_atomic_load(abort_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                        aborted = true;
                    }
                    if (aborted) break;
                    __builtin_amdgcn_s_sleep(2);
                    continue;
                }
                uint32_t a = lane < p ? (uint32_t)d : 0u;  // (a tile's own count fits 17 bits)
#pragma unroll
                for (int s2 = 32; s2 >= 1; s2 >>= 1) a += __shfl_xor(a, s2, kWave);
                prefix += a;
                if (mp) {
                    const uint32_t lo = __shfl((uint32_t)d, (int)p, kWave), hi = __shfl((uint32_t)(d >> 32), (int)p, kWave);
                    prefix += (((uint64_t)hi << 32) | lo) & kDescVal;
                    break;
                }
                idx -= kWave;
            }
            if (!aborted && !stall && lane == 0) __hip_atomic_store(desc + tile, kDescP | (prefix + tile_count), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) {
            s_prefix = prefix;
            s_abort = aborted ? 1u : 0u;
            const uint64_t incl = prefix + tile_count;
            // (the range's total through an atomic MAX over a word the launcher cleared: a tile that gave up leaves the sentinel -- the
            //  largest value -- whichever of the two stores lands last; tiles behind a tile that gave up still finish, on its count)
            if (aborted) {
                atomicMax(reinterpret_cast<unsigned long long*>(pref + blk1), (unsigned long long)kFrameAborted);
            } else {
                if (tile == (blk1 - 1u) / kTileWaves)  // the delimiters up to the end of this launch's range
                    atomicMax(reinterpret_cast<unsigned long long*>(pref + blk1), (unsigned long long)incl);
                if (tile == 0) offsets[0] = 0;
                if ((whole & kWholeStream) && tile == (nblk - 1u) / kTileWaves && incl + 1 <= cap) offsets[incl + 1] = nbytes;  // a final unterminated frame
            }
        }
    }
    __syncthreads();
    if (s_abort) return;
    if (mine == 0 && any_err == 0) return;
    uint64_t rank = s_prefix + wave_off + ex;  // delimiters before this lane's first byte
    const uint64_t lane_base = base + (uint64_t)lane * 256u;
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
// classic != 0: the three-kernel form (the fall-back of a one-pass launch that reported kFrameAborted, and FG_LO_FRAME_CLASSIC).
// *d_total_out: the device word that will hold the delimiter count -- or FG_FRAME_ABORTED (~0), after which the caller launches again
// with classic = 1 (same arguments; everything is rewritten).
static inline uint32_t* frame_abort_word(uint8_t* scratch, uint64_t nblk) {
    return reinterpret_cast<uint32_t*>(scratch + nblk * 4096u + ((nblk * 4u + 255u) & ~255ull) + (nblk + 1u) * 8u + 64u);
}
extern "C" int fg_launch_frame(const uint8_t* d_bytes, uint64_t nbytes, uint32_t delim, uint8_t* scratch, uint64_t* d_offsets,
                               uint8_t* d_bad, uint64_t cap, uint64_t** d_total_out, hipStream_t stream, int classic) {
    const uint64_t nblk = frame_blocks(nbytes);
    if (nblk > 0x7FFFFFFFull) return -1;
    uint32_t* masks = reinterpret_cast<uint32_t*>(scratch);
    uint32_t* counts = reinterpret_cast<uint32_t*>(scratch + nblk * 4096u);
    uint64_t* pref = reinterpret_cast<uint64_t*>(scratch + nblk * 4096u + ((nblk * 4u + 255u) & ~255ull));
    const uint32_t pat = delim * 0x01010101u;
    (void)hipMemsetAsync(d_bad, 0, cap, stream);
    *d_total_out = pref + nblk;
    if (classic != 1) {  // (classic == 2: the one-pass form with its self-test stall -- it WILL report FG_FRAME_ABORTED on a stream of three tiles or more)
        uint64_t* desc = reinterpret_cast<uint64_t*>(scratch);  // (where the classic form keeps its masks)
        (void)hipMemsetAsync(desc, 0, nblk * 8u, stream);
        (void)hipMemsetAsync(frame_abort_word(scratch, nblk), 0, 4, stream);
        (void)hipMemsetAsync(pref + nblk, 0, 8, stream);
        hipLaunchKernelGGL(fg::k_frame_onepass<false>, dim3((uint32_t)((nblk + fg::kTileWaves - 1) / fg::kTileWaves)), dim3(fg::kWave * fg::kTileWaves), 0,
                           stream, d_bytes, nbytes, pat, desc, pref, frame_abort_word(scratch, nblk), d_offsets, d_bad, cap, (uint64_t)0, nblk, nblk,
                           fg::kWholeStream | (classic == 2 ? fg::kSelfTestStall : 0u), (uint8_t*)nullptr);
        return (int)hipGetLastError();
    }
    hipLaunchKernelGGL(fg::k_frame_scan<false>, dim3((uint32_t)nblk), dim3(fg::kWave), 0, stream, d_bytes, nbytes, pat, masks, counts, (uint64_t)0,
                       (uint8_t*)nullptr);
    hipLaunchKernelGGL(fg::k_frame_prefix, dim3(1), dim3(1024), 0, stream, counts, nblk, pref, 0u);
    hipLaunchKernelGGL(fg::k_frame_emit, dim3((uint32_t)nblk), dim3(fg::kWave), 0, stream, masks, pref, nbytes, nblk, d_offsets,
                       d_bad, cap, (uint64_t)0, 1u);
    return (int)hipGetLastError();
}

// One SLICE of the stream: the blocks [blk0, blk1) (slice boundaries are multiples of the 16 KiB block; the last slice ends at
// frame_blocks(nbytes)), whose bytes -- and everything before them -- are in d_bytes.  Continues the delimiter ranks where the
// slice before stopped; *d_total_out = the device word that holds the delimiters up to the end of this slice.  d_bad must have
// been cleared for the whole batch beforehand; offsets[total + 1] of a final unterminated frame is the caller's business.
// Slices of one stream are launched in order, all classic or all one-pass (the first slice clears the descriptors of the whole stream).
extern "C" uint64_t fg_frame_block_bytes(void) { return fg::kFrameBlock; }
extern "C" uint64_t fg_frame_slice_align(void) { return (uint64_t)fg::kFrameBlock * fg::kTileWaves; }  // slice boundaries: whole tiles
// src != null: the slice's bytes are read from `src` (the device view of the caller's pinned buffer, same offsets) and stored to d_bytes
// by the scan itself -- no upload has to have happened.
extern "C" int fg_launch_frame_slice(const uint8_t* d_bytes, uint64_t nbytes, uint32_t delim, uint8_t* scratch, uint64_t* d_offsets,
                                     uint8_t* d_bad, uint64_t cap, uint64_t blk0, uint64_t blk1, uint64_t** d_total_out,
                                     hipStream_t stream, const uint8_t* src, int classic) {
    const uint64_t nblk = frame_blocks(nbytes);
    if (nblk > 0x7FFFFFFFull || blk1 > nblk || blk0 >= blk1) return -1;
    uint32_t* masks = reinterpret_cast<uint32_t*>(scratch);
    uint32_t* counts = reinterpret_cast<uint32_t*>(scratch + nblk * 4096u);
    uint64_t* pref = reinterpret_cast<uint64_t*>(scratch + nblk * 4096u + ((nblk * 4u + 255u) & ~255ull));
    const uint32_t pat = delim * 0x01010101u;
    const uint32_t nb = (uint32_t)(blk1 - blk0);
    *d_total_out = pref + blk1;
    if (classic != 1) {
        const uint32_t stall = (classic == 2 && blk0 == 0) ? fg::kSelfTestStall : 0u;  // (self-test: the stream's first slice stalls)
        uint64_t* desc = reinterpret_cast<uint64_t*>(scratch);
        if (blk0 == 0) {
            (void)hipMemsetAsync(desc, 0, nblk * 8u, stream);
            (void)hipMemsetAsync(frame_abort_word(scratch, nblk), 0, 4, stream);
        }
        // (tiles of kTileWaves blocks: a slice that is not the stream's first starts at a multiple of fg_frame_slice_align())
        if (blk0 % fg::kTileWaves) return -1;
        (void)hipMemsetAsync(pref + blk1, 0, 8, stream);
        const uint64_t tile0 = blk0 / fg::kTileWaves;
        const uint32_t nt = (uint32_t)((blk1 + fg::kTileWaves - 1) / fg::kTileWaves - tile0);
        if (src)
            hipLaunchKernelGGL(fg::k_frame_onepass<true>, dim3(nt), dim3(fg::kWave * fg::kTileWaves), 0, stream, src, nbytes, pat, desc, pref,
                               frame_abort_word(scratch, nblk), d_offsets, d_bad, cap, tile0, blk1, nblk, stall, const_cast<uint8_t*>(d_bytes));
        else
            hipLaunchKernelGGL(fg::k_frame_onepass<false>, dim3(nt), dim3(fg::kWave * fg::kTileWaves), 0, stream, d_bytes, nbytes, pat, desc, pref,
                               frame_abort_word(scratch, nblk), d_offsets, d_bad, cap, tile0, blk1, nblk, stall, (uint8_t*)nullptr);
        return (int)hipGetLastError();
    }
    if (src)
        hipLaunchKernelGGL(fg::k_frame_scan<true>, dim3(nb), dim3(fg::kWave), 0, stream, src, nbytes, pat, masks, counts, blk0,
                           const_cast<uint8_t*>(d_bytes));
    else
        hipLaunchKernelGGL(fg::k_frame_scan<false>, dim3(nb), dim3(fg::kWave), 0, stream, d_bytes, nbytes, pat, masks, counts, blk0, (uint8_t*)nullptr);
    hipLaunchKernelGGL(fg::k_frame_prefix, dim3(1), dim3(1024), 0, stream, counts + blk0, (uint64_t)nb, pref + blk0, blk0 ? 1u : 0u);
    hipLaunchKernelGGL(fg::k_frame_emit, dim3(nb), dim3(fg::kWave), 0, stream, masks, pref, nbytes, nblk, d_offsets, d_bad, cap, blk0, 0u);
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

