// fg_plan_policy.hpp -- the HOST-side policies of a decode launch that are pure arithmetic: how the lines of a batch are cut into chunks
// (and whether chunks beyond a wave's first are drawn from the launch's ticket counter), and how many entry slots a wave reserves at a
// time.  No HIP dependency: plan_launch (fg_pipeline.hpp) calls these, and tests/test_plan_policy_cpu.py sweeps them on the CPU.
#pragma once
#include <stdint.h>

#include "../../include/fg_hip.h"

#if defined(__HIPCC__)
#define FG_POLICY_HD __host__ __device__
#else
#define FG_POLICY_HD
#endif

namespace fg {

// The TAPER of a ticket launch: chunks are `chunk` lines up to chunk index taper[0], half that up to taper[1], a quarter up to
// taper[2], an eighth from there to the end of the batch (kNoTaper = that level is not used; taper[0] <= taper[1] <= taper[2]).
// Waves that draw chunks of C lines finish up to one chunk's time apart, so the LAST chunk per wave's worth of lines is dealt out in
// smaller pieces (guided self-scheduling, at most three levels).  Measured, one box, every setting alternated
// (profiles/r05u_chunk_taper_sweep.log): ONE level is what pays -- structured data 4 M lines, chunks of 128: 1956 M lines/s without,
// 1990 with one level, 1981 / 1979 with two / three; 16 M lines, chunks of 512: 2042 -> 2080; LTSV 4 M lines 3813 -> 3941; the
// long-tail corpus 1366 -> 1396 -- and the HBM-bound headline kernel gains nothing from it (14.75-14.88 G without, 14.86-14.88 with
// one level, 14.63 with two: its last level would be chunks of ONE group, which that kernel cannot afford).  Deeper levels are a
// tuning option (FG_LO_TAPER_1 / _2).
constexpr uint32_t kNoTaper = 0xFFFFFFFFu;

// chunk index -> its lines [lo, hi) (lo == hi == n: beyond the batch).  Shared by the kernels (persistent_loop) and the host's count.
FG_POLICY_HD inline void chunk_range(uint64_t c, uint64_t chunk, uint32_t t0, uint32_t t1, uint32_t t2, uint64_t n, uint64_t* lo,
                                     uint64_t* hi) {
    uint64_t base = 0, first = 0, sz = chunk;
    if (c >= t0) {
        base = (uint64_t)t0 * sz, first = t0, sz >>= 1;
        if (c >= t1) {
            base += (uint64_t)(t1 - t0) * sz, first = t1, sz >>= 1;
            if (c >= t2) base += (uint64_t)(t2 - t1) * sz, first = t2, sz >>= 1;
        }
    }
    const uint64_t l = base + (c - first) * sz;
    *lo = l < n ? l : n;
    *hi = l + sz < n ? l + sz : n;
}

// chunks a batch of n lines makes under a taper (the first index whose range is empty)
inline uint64_t count_chunks(uint64_t chunk, const uint32_t t[3], uint64_t n) {
    uint64_t base = 0, first = 0, sz = chunk;
    for (int j = 0; j < 3; ++j) {
        if (t[j] == kNoTaper) break;
        const uint64_t upto = base + (uint64_t)(t[j] - first) * sz;
        if (upto >= n) break;
        base = upto, first = t[j], sz >>= 1;
    }
    return first + (n - base + sz - 1) / sz;
}

struct ChunkPlan {
    uint64_t chunk = 1;    // lines per chunk
    uint64_t chunks = 0;   // chunks of the batch (ceil(n / chunk) without a taper)
    uint32_t blocks = 0;   // persistent waves (<= chunks)
    bool tickets = false;  // chunks beyond a wave's first are drawn from the launch's ticket counter (else: round-robin)
    uint32_t taper[3] = {kNoTaper, kNoTaper, kNoTaper};  // (ticket launches only)
};

//   n           lines of the launch (>= 1)
//   blocks      waves the occupancy allows (>= 1)
//   L           lines per group at most (the wave width or the format's cap)
//   g           lines an AVERAGE group holds (groups are cut by bytes), 1 .. L
//   full        the format's chunk for large batches (256 / 512 / its own choice)
//   ticket_from chunks of `full` lines per wave from which tickets are drawn
//   lo          the caller's launch options (chunk_lines, FG_LO_STATIC_CHUNKS, FG_LO_NO_TAPER / _TAPER_1 / _TAPER_2)
//   taper       levels of the taper at the end of a ticket launch, 0 .. 3 (the format's choice)
inline ChunkPlan plan_chunks(uint64_t n, uint64_t blocks, uint32_t L, uint64_t g, uint64_t full, uint32_t ticket_from, const fg_launch_opts& lo,
                             uint32_t taper = 1) {
    ChunkPlan p;
    if (blocks < 1u) blocks = 1u;
    if (g < 1u) g = 1u;
    if (g > L) g = L;
    if (full < 1u) full = 1u;
    const bool dynamic = !(lo.flags & FG_LO_STATIC_CHUNKS);
    const bool named = lo.chunk_lines >= (dynamic ? 1u : L) && lo.chunk_lines <= 65536u;
    const uint64_t per_wave = (n + blocks - 1) / blocks;
    uint64_t chunk;
    bool tickets = false;
    if (named) {
        chunk = lo.chunk_lines;  // (tuning: taken as it is, unless the batch is too small for two of them per wave)
        if (n < blocks * 2u * chunk) {
            chunk = per_wave;
            if (chunk < (dynamic ? g : (uint64_t)L)) chunk = dynamic ? g : (uint64_t)L;
        }
        tickets = dynamic;  // (a named chunk size under dynamic dispatch always draws: tests, tuning)
    } else if (dynamic && per_wave >= (uint64_t)ticket_from * full) {
        // Chunks are DRAWN (persistent_loop's ticket): a wave that meets slow lines or sits on a slow XCD draws fewer -- the balance is
        // the dispatch's business, so the chunk is simply `full` lines: its last group is as short as it comes out and costs these
        // latency-bound kernels what a full one does (1 group in 4 .. 50), and every ticket is an atomic on one word, of which the
        // chip serves a few dozen per microsecond -- chunks of ONE group (8192 tickets for 512 K lines of the headline corpus) doubled
        // the kernel's time (profiles/r05b_small_ab.log: 123 vs 65 us).  A batch with fewer than `ticket_from` such chunks per wave
        // takes the equal shares below and draws nothing: two to four for the compute-bound kernels (GELF gains 13 % from six chunks per
        // wave on, LTSV 5 %; structured data and GELF lose 1-4 % at three: profiles/r05v_policy_ab.log), twenty for the HBM-bound
        // headline kernel, whose first round of tickets -- 1792 waves start at the same
        // moment -- arrives as a burst on one word, and an atomic that takes 20 us to come back holds the wave's window loads up behind
        // it (vmcnt retires in order): +25 us at 1 M, 2 M and 4 M lines alike, +12 % throughput at 16 M, +18 % at 40 M
        // (profiles/r05d_small_cfg2_big.log, r05e_small.log, r05d_sweep_cfg2_40M.log).
        chunk = full;
        tickets = true;
    } else if (dynamic) {
        // ONE chunk per wave, the wave's whole share: a single ragged group per wave, and no floor of L lines -- a small batch is cut
        // down to one average group (less a sixteenth: a unit that USUALLY is one group) per wave (16 K structured-data lines, 21 to the
        // group, ran as 256 waves of three groups each on a grid of 2048: 84 us where 47 do).  Same box, alternated, against the k equal
        // chunks of <= `full` lines of rounds 3-4 (profiles/r05f_small_cfg2_chunks.log, r05f_sweep_4M_chunks.log): headline corpus 1 M
        // lines 97 vs 104 us, 4 M 340 vs 335; structured data 256 K / 512 K / 4 M lines 194 / 328 us / 1.91 G against 195 / 330 us / 1.89 G;
        // GELF and LTSV alike.  (The sweep-front-to-back argument for many chunks per wave is a large batch's: that regime draws tickets.)
        chunk = per_wave;
        const uint64_t unit = g >= L ? g : (g * 15u / 16u ? g * 15u / 16u : 1u);
        if (chunk < unit) chunk = unit;
    } else {
        // round-robin (rounds 3-4, FG_LO_STATIC_CHUNKS): every wave the SAME number of chunks -- k = the chunks per wave that keeps a
        // chunk at or below `full`, chunk = n / (waves * k).  (With chunks of exactly `full` lines a batch of 1.9 chunks per wave left a
        // tenth of the grid with half the work of the rest: profiles/r04z3_sweep_cfg4.log, r04z5_sweep_cfg5.log.)
        const uint64_t k = (per_wave + full - 1) / full;
        chunk = (per_wave + (k ? k : 1) - 1) / (k ? k : 1);
        if (chunk < L) chunk = L;
    }
    if (chunk < 1u) chunk = 1u;
    p.chunk = chunk;
    p.chunks = (n + chunk - 1) / chunk;
    if (lo.flags & (FG_LO_TAPER_1 | FG_LO_TAPER_2)) taper = ((lo.flags & FG_LO_TAPER_1) ? 1u : 0u) + ((lo.flags & FG_LO_TAPER_2) ? 2u : 0u);
    if (taper > 3u) taper = 3u;
    if (tickets && taper && !(lo.flags & FG_LO_NO_TAPER)) {
        // the tail: one chunk per wave's worth of lines, at most half the batch; levels down to one average group (a chunk below
        // that is a group short of lines at a full group's latency)
        uint64_t m = n / (2u * chunk);
        if (m > blocks) m = blocks;
        uint32_t levels = 0;
        while (levels < taper && (chunk >> (levels + 1u)) >= g) ++levels;
        if (m >= 1u && levels >= 1u && p.chunks + 4u * m < 0xFFFFFFF0ull) {
            const uint64_t t0 = (n - m * chunk) / chunk;  // full chunks before the tail
            p.taper[0] = (uint32_t)t0;
            if (levels >= 2u) p.taper[1] = (uint32_t)(t0 + m);
            if (levels >= 3u) p.taper[2] = (uint32_t)(t0 + 2u * m);
            p.chunks = count_chunks(chunk, p.taper, n);
        }
    }
    p.blocks = (uint32_t)(blocks > p.chunks ? p.chunks : blocks);
    p.tickets = tickets;
    return p;
}

// Entry slots a wave reserves from the table's counter at a time (DevTables::alloc_chunk; wv::wave_alloc).  Every reservation is an
// atomic on ONE word, and that word's channel serves only a few dozen of them per microsecond (rounds 3-4 and profiles/r05b_*: GELF,
// one request per 8-line tile, took 290 us for 64 K lines and 915 us for 256 K where 512 K take 400 us -- small launches fell under
// the 256-slot floor of wv::alloc_chunk_for into EXACT reservations).  So: the table's share per wave (1/16 of it over all waves, at
// most 4096 slots) capped by eight slots per line the wave will see in this launch -- but never below 256 slots (a floor of 64 is one
// 8-line GELF tile's worth: still an atomic per tile; 256: 340 -> 184 us for 64 K lines, 896 -> 303 us for 256 K,
// profiles/r05c_small_gelf_opts.log); exact reservations (0) only where a quarter of the table does not hold 256 slots per wave (a
// caller that sized the table tightly must not see FG_ST_OVERFLOW because of slots parked in chunks).  What a wave strands
// is the rest of its last chunk: half a chunk on average, never written and -- on the zero-copy host paths, which write the tables
// across the link themselves -- never moved.
// shares: launches that reserve from the SAME table (the sliced host paths decode one batch as 6 .. 64 launches into one table sized
// from the input bytes): the per-launch rule above bounds what ONE launch strands, so the table's share per wave is that of a
// 1 / shares part of it -- else the slices' stranded chunks add up past the capacity, the batch reports FG_ERR_UNSUPPORTED and is decoded
// again on the one-piece path: a cliff on structured-data-heavy batches.
inline uint32_t entry_chunk(uint64_t ent_cap, uint32_t blocks, uint64_t n, const fg_launch_opts& lo, uint32_t shares = 1) {
    if (lo.ent_chunk == 1u) return 0u;
    if (lo.ent_chunk >= 2u) return lo.ent_chunk;
    const uint64_t waves = blocks ? blocks : 1u;
    const uint64_t budget = ent_cap / (shares ? shares : 1u);
    uint64_t c = budget / (16ull * waves);
    if (c > 4096u) c = 4096u;
    if (budget / (4ull * waves) < 256u) return 0u;  // (a quarter of the table's share stranded at worst, an eighth on average)
    if (c < 256u) c = 256u;
    const uint64_t per_wave = (n + waves - 1u) / waves * 8ull;
    if (c > per_wave) c = per_wave;
    if (c < 256u) c = 256u;
    return (uint32_t)c;
}

}  // namespace fg
