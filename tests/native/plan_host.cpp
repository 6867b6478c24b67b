// C wrappers around flowgger_amd/csrc/fg_plan_policy.hpp for tests/test_plan_policy_cpu.py (the launch policies are host arithmetic)
#include "../../flowgger_amd/csrc/fg_plan_policy.hpp"

extern "C" {
// out[0] chunk, out[1] chunks, out[2] blocks, out[3] tickets, out[4..6] taper
void fgp_plan_chunks(uint64_t n, uint64_t blocks, uint32_t L, uint64_t g, uint64_t full, uint32_t ticket_from, uint32_t flags, uint32_t chunk_lines,
                     uint32_t taper, uint64_t* out) {
    fg_launch_opts lo{};
    lo.flags = flags;
    lo.chunk_lines = chunk_lines;
    const fg::ChunkPlan p = fg::plan_chunks(n, blocks, L, g, full, ticket_from, lo, taper);
    out[0] = p.chunk;
    out[1] = p.chunks;
    out[2] = p.blocks;
    out[3] = p.tickets ? 1 : 0;
    for (int j = 0; j < 3; ++j) out[4 + j] = p.taper[j];
}
void fgp_chunk_range(uint64_t c, uint64_t chunk, uint32_t t0, uint32_t t1, uint32_t t2, uint64_t n, uint64_t* lo_hi) {
    fg::chunk_range(c, chunk, t0, t1, t2, n, &lo_hi[0], &lo_hi[1]);
}
uint32_t fgp_entry_chunk(uint64_t ent_cap, uint32_t blocks, uint64_t n, uint32_t ent_chunk) {
    fg_launch_opts lo{};
    lo.ent_chunk = ent_chunk;
    return fg::entry_chunk(ent_cap, blocks, n, lo);
}
uint32_t fgp_entry_chunk_shared(uint64_t ent_cap, uint32_t blocks, uint64_t n, uint32_t ent_chunk, uint32_t shares) {
    fg_launch_opts lo{};
    lo.ent_chunk = ent_chunk;
    return fg::entry_chunk(ent_cap, blocks, n, lo, shares);
}
}
