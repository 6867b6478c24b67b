// fg_merge.hip -- the ORDERED MERGE of tagged sub-batches on the device (BASELINE configuration 5: the reference has one
// decoder per input, flowgger/mod.rs:413-422, so a mixed stream is decoded as sub-batches split off by format; the order contract:
// handle_line runs in input order per connection, src/flowgger/splitter/line_splitter.rs:17-54).
//
// fg_merge_tables (fg_gather.cpp) does this on the HOST after every sub-batch's tables have crossed the link: 1.1 GB of rows and
// entries re-threaded by the CPU at ~32 GB/s -- 36 of the 56 ms `gather_ms` of a 4 M-line mixed batch.  Here the rows go back to their
// arrival positions while the tables are still in HBM (0.3 ms of streaming), and ONE merged table crosses the link.
//   rows     lane = row j of part k: the ten fixed columns to position index[k][j]; ent_first rebased onto the merged entry table
//   entries  the parts' entry columns, [0, used_k) of each, behind one another (a part's slices keep their relative positions: slots no
//            line refers to travel along, as in fg_gather_tables)
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
    uint8_t* src_part;
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
    uint64_t base = 0;
    for (uint32_t q = 0; q < k; ++q) base += used_of(a.part[q]);
    const uint64_t* ix = a.index[k];
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
        const uint32_t cnt = p.ent_count[j];
        a.out.ent_count[i] = cnt;
        a.out.ent_first[i] = cnt ? (uint32_t)(p.ent_first[j] + base) : 0u;
        if (a.src_part) a.src_part[i] = (uint8_t)k;
    }
    if (k == 0 && blockIdx.x == 0 && threadIdx.x == 0) {
        uint64_t total = 0;
        for (uint32_t q = 0; q < a.g; ++q) total += used_of(a.part[q]);
        *reinterpret_cast<unsigned long long*>(a.out.ent_used) = total;
    }
}

__global__ __launch_bounds__(256) void k_merge_entries(MergeArgs a) {
    const uint32_t k = blockIdx.y;
    const fg_tables& p = a.part[k];
    uint64_t base = 0;
    for (uint32_t q = 0; q < k; ++q) base += used_of(a.part[q]);
    const uint64_t used = used_of(p);
    if (base + used > a.out.ent_cap) return;  // (cannot happen behind the host entry point's size check; never write past the table)
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < used; e += (uint64_t)gridDim.x * blockDim.x) {
        a.out.ent_name[base + e] = p.ent_name[e];
        a.out.ent_val[base + e] = p.ent_val[e];
        a.out.ent_type[base + e] = p.ent_type[e];
        a.out.ent_flags[base + e] = p.ent_flags[e];
    }
}

}  // namespace fg

// parts / index / out: device-addressable; asynchronous on `stream`.  max_rows / max_entries: the largest part (grid sizing only).
extern "C" int fg_launch_merge_device(const fg_tables* parts, uint32_t g, const uint64_t* const* d_index, const fg_tables* out,
                                      uint8_t* d_src_part, uint64_t max_rows, uint64_t max_entries, hipStream_t stream) {
    if (g == 0 || g > fg::kMergeParts) return -1;
    fg::MergeArgs a{};
    for (uint32_t k = 0; k < g; ++k) {
        a.part[k] = parts[k];
        a.index[k] = d_index[k];
    }
    a.out = *out;
    a.src_part = d_src_part;
    a.g = g;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1)
        return -1;
    auto blocks_for = [&](uint64_t items) {
        uint64_t b = (items + 255u) / 256u;
        const uint64_t cap = (uint64_t)cus * 16u;
        if (b > cap) b = cap;
        return (uint32_t)(b ? b : 1u);
    };
    hipLaunchKernelGGL(fg::k_merge_rows, dim3(blocks_for(max_rows), g), dim3(256), 0, stream, a);
    if (max_entries) hipLaunchKernelGGL(fg::k_merge_entries, dim3(blocks_for(max_entries), g), dim3(256), 0, stream, a);
    return (int)hipGetLastError();
}
