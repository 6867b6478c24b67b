// fg_merge.hip -- the ORDERED MERGE of tagged sub-batches on the device (BASELINE configuration 5: the reference has one
// decoder per input, flowgger/mod.rs:413-422, so a mixed stream is decoded as sub-batches split off by format; the order contract:
// handle_line runs in input order per connection, src/flowgger/splitter/line_splitter.rs:17-54).
//
// fg_merge_tables (fg_gather.cpp) does this on the HOST after every sub-batch's tables have crossed the link: 1.1 GB of rows and
// entries re-threaded by the CPU at ~32 GB/s -- 36 of the 56 ms `gather_ms` of a 4 M-line mixed batch.  Here the rows go back to their
// arrival positions while the tables are still in HBM (0.3 ms of streaming), and ONE merged table crosses the link.
//   rows     lane = row j of part k: the ten fixed columns to position index[k][j]
//   entries  DENSE and in ARRIVAL ORDER since round 5: an exclusive scan over ent_count in arrival order gives every
//            row its slice of the merged entry table, and a wave copies the entries of its 64 rows to 64 consecutive output slots per
//            store instruction.  (Round 4 laid the parts' entry columns behind one another, reserved-but-unreferenced slots included --
//            the waves' stranded reservations travelled across the link: 13.29 M entries where the lines own 11.53 M.)
// Plain streaming copies: HBM-bound, nothing to tile.  No reference analogue beyond the order contract above.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fg_hip.h"

namespace fg {

constexpr uint32_t kMergeParts = 8;
struct MergeArgs {
    fg_tables part[kMergeParts];
    const uint64_t* index[kMergeParts];
    fg_tables out;
    uint8_t* src_part;      // [out.n] which part a merged row came from (the caller's array, or scratch)
    uint64_t* dense;        // [out.n + 1] scratch: exclusive scan of ent_count in arrival order
    uint64_t* block_sums;   // [ceil(out.n / 64)] scratch
    uint32_t g;
};

// entries of part k that exist (its counter may have run past its capacity: those rows carry FG_ST_OVERFLOW)
__device__ __forceinline__ uint64_t used_of(const fg_tables& p) {
    const uint64_t u = *reinterpret_cast<const unsigned long long*>(p.ent_used);
    return u < p.ent_cap ? u : p.ent_cap;
}

__global__ __launch_bounds__(256) void k_merge_rows(MergeArgs a) {
    const uint32_t k = blockIdx.y;
    const fg_tables& p = a.part[k];
    const uint64_t* ix = a.index[k];
    const uint64_t used = used_of(p);
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < p.n; j += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t i = ix[j];
        if (i >= a.out.n) continue;  // (an index outside the merged table: the host entry point has checked the sizes, not every index)
        a.out.meta[i] = p.meta[j];
        a.out.ts[i] = p.ts[j];
        a.out.hostname[i] = p.hostname[j];
        a.out.appname[i] = p.appname[j];
        a.out.procid[i] = p.procid[j];
        a.out.msgid[i] = p.msgid[j];
        a.out.msg[i] = p.msg[j];
        a.out.full_msg[i] = p.full_msg[j];
        // (a slice that does not lie inside the part's entries -- a row of a table that overflowed -- owns nothing in the merged table)
        uint32_t cnt = p.ent_count[j];
        const uint32_t first = p.ent_first[j];
        if ((uint64_t)first + cnt > used) cnt = 0;
        a.out.ent_count[i] = cnt;
        a.out.ent_first[i] = first;  // part-LOCAL until k_merge_entries has moved the slice
        a.src_part[i] = (uint8_t)k;
    }
}

// entries owned by the rows of every 64-row block of the merged table (arrival order)
__global__ __launch_bounds__(64) void k_merge_block_sums(MergeArgs a) {
    const uint64_t i = (uint64_t)blockIdx.x * 64u + threadIdx.x;
    uint64_t c = i < a.out.n ? a.out.ent_count[i] : 0u;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
    if (threadIdx.x == 0) a.block_sums[blockIdx.x] = c;
}

// One wave per 64 merged rows: their entries occupy the dense range [dense[g0], dense[g1]) of the merged table; lane e of a trip
// writes output slot dense[g0] + e -- consecutive lanes, consecutive slots -- from wherever its row's slice lies in the row's part.
__global__ __launch_bounds__(64) void k_merge_entries(MergeArgs a) {
    __shared__ uint64_t s_ds[65];
    __shared__ uint32_t s_src[64];
    __shared__ uint8_t s_k[64];
    __shared__ const fg_span* s_name[kMergeParts];
    __shared__ const uint64_t* s_val[kMergeParts];
    __shared__ const uint8_t* s_type[kMergeParts];
    __shared__ const uint8_t* s_flags[kMergeParts];
    const uint32_t lane = threadIdx.x;
    const uint64_t g0 = (uint64_t)blockIdx.x * 64u;
    const uint32_t rows = a.out.n - g0 < 64u ? (uint32_t)(a.out.n - g0) : 64u;
    if (lane < a.g) {  // (the parts' column pointers by part number, out of LDS: a lane's part differs from its neighbour's)
        s_name[lane] = a.part[lane].ent_name;
        s_val[lane] = a.part[lane].ent_val;
        s_type[lane] = a.part[lane].ent_type;
        s_flags[lane] = a.part[lane].ent_flags;
    }
    const uint64_t i = g0 + lane;
    if (lane < rows) {
        const uint64_t ds = a.dense[i];
        s_ds[lane] = ds;
        s_src[lane] = a.out.ent_first[i];
        s_k[lane] = a.src_part[i];
        a.out.ent_first[i] = a.out.ent_count[i] ? (uint32_t)ds : 0u;
    }
    if (lane == 0) s_ds[rows] = a.dense[g0 + rows];
    // (ent_used = what the merged rows need; above ent_cap -- or above what ent_first can address -- the caller sees the overflow there)
    if (blockIdx.x == 0 && lane == 0) *reinterpret_cast<unsigned long long*>(a.out.ent_used) = a.dense[a.out.n];
    __syncthreads();
    const uint64_t d0 = s_ds[0];
    const uint32_t total = (uint32_t)(s_ds[rows] - d0);
    for (uint32_t e = lane; e < total; e += 64u) {
        const uint64_t t = d0 + e;
        // the row that owns slot t: the last one whose slice starts at or before it (rows without entries never do: upper bound)
        uint32_t lo = 0, hi = rows;
        while (hi - lo > 1u) {
            const uint32_t mid = (lo + hi) >> 1;
            if (s_ds[mid] <= t) lo = mid; else hi = mid;
        }
        const uint32_t k = s_k[lo];
        const uint64_t src = (uint64_t)s_src[lo] + (t - s_ds[lo]);
        // (the dense total is bounded by the parts' used counts only while the index is the caller's contract -- every position once, no two
        //  rows sharing a slice; an index that repeats a row must not become a write past the table: ADVICE r5)
        if (t >= a.out.ent_cap) continue;
        a.out.ent_name[t] = s_name[k][src];
        a.out.ent_val[t] = s_val[k][src];
        a.out.ent_type[t] = s_type[k][src];
        a.out.ent_flags[t] = s_flags[k][src];
    }
}

}  // namespace fg

extern "C" int fg_launch_encode_scan(const uint32_t* d_sizes, uint64_t* d_block_sums, uint64_t n, uint64_t* d_out_offsets, uint64_t base,
                                     hipStream_t stream);  // (fg_encode.hip: per-64 sums -> exclusive scan per row; out[n] = the total)
// scratch the merge needs for `rows` merged rows: dense[rows + 1] u64 | block sums | src_part (when the caller passes none)
extern "C" uint64_t fg_merge_scratch_bytes(uint64_t rows) { return (rows + 1u) * 8u + (rows / 64u + 2u) * 8u + rows + 64u; }

// parts / index / out: device-addressable; asynchronous on `stream`.  max_rows: the largest part (grid sizing only).
// scratch: fg_merge_scratch_bytes(out->n) bytes of device memory, 8-byte aligned.
extern "C" int fg_launch_merge_device(const fg_tables* parts, uint32_t g, const uint64_t* const* d_index, const fg_tables* out,
                                      uint8_t* d_src_part, uint64_t max_rows, uint8_t* scratch, hipStream_t stream) {
    if (g == 0 || g > fg::kMergeParts || !scratch) return -1;
    fg::MergeArgs a{};
    for (uint32_t k = 0; k < g; ++k) {
        a.part[k] = parts[k];
        a.index[k] = d_index[k];
    }
    a.out = *out;
    a.g = g;
    const uint64_t n = out->n, nb = (n + 63u) / 64u;
    a.dense = reinterpret_cast<uint64_t*>(scratch);
    a.block_sums = a.dense + (n + 1u);
    a.src_part = d_src_part ? d_src_part : reinterpret_cast<uint8_t*>(a.block_sums + (n / 64u + 2u));
    if (n == 0) {
        (void)hipMemsetAsync(out->ent_used, 0, 8, stream);
        return (int)hipGetLastError();
    }
    if (nb > 0x7FFFFFFFull) return -1;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1)
        return -1;
    auto blocks_for = [&](uint64_t items) {
        uint64_t b = (items + 255u) / 256u;
        const uint64_t cap = (uint64_t)cus * 16u;
        if (b > cap) b = cap;
        return (uint32_t)(b ? b : 1u);
    };
    // (rows no index names keep whatever the table held: the counts of such rows must not reach the scan)
    (void)hipMemsetAsync(out->ent_count, 0, n * 4u, stream);
    hipLaunchKernelGGL(fg::k_merge_rows, dim3(blocks_for(max_rows), g), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(fg::k_merge_block_sums, dim3((uint32_t)nb), dim3(64), 0, stream, a);
    if (fg_launch_encode_scan(out->ent_count, a.block_sums, n, a.dense, 0ull, stream) != 0) return -1;
    hipLaunchKernelGGL(fg::k_merge_entries, dim3((uint32_t)nb), dim3(64), 0, stream, a);
    return (int)hipGetLastError();
}
