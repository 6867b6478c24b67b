// fg_fused_plan.hpp -- HOST-side geometry of a fused (frame + decode) launch: pure arithmetic, no HIP and no wave header, shared by the
// kernel launchers (fg_rfc5424.hip, fg_ltsv.hip, fg_gelf.hip), the C ABI (fg_capi.cpp sizes the launch's scratch from it) and the CPU
// tests.  The device side is fg_fused.hpp; the tile-level framing logic fg_fuse.hpp.
#pragma once
#include <stdint.h>

#include "../../include/fg_hip.h"

namespace fg {
namespace fuse {

constexpr uint32_t kPreBytes = 16;  // (= fuse::kPre)

// S: as many bytes as `lines` average lines hold, less an eighth (a tile that holds more lines than a pass of stage B takes costs a
// second pass over the same tile), a multiple of 256 (of 16 for small tiles); look: one average line rounded up to 64, 64 .. 2048 --
// the tile's last line ends within it most of the time, else the wave reads on (fuse::forward_scan) and the line is parsed from global
// memory.  Both bounded by the LDS tile: kPre + S + look + 16 <= tile_cap.
struct TilePlan { uint32_t S, look; };
inline TilePlan plan_tile(uint64_t avg_len, uint32_t lines, uint32_t tile_cap, uint32_t look_override = 0) {
    if (avg_len < 16u) avg_len = 16u;
    uint64_t look = (avg_len + 63u) & ~63ull;
    if (look > 2048u) look = 2048u;
    if (look_override) look = (look_override + 15u) & ~15u;  // (tuning: fg_launch_opts.fused_look)
    if (look + 1024u > tile_cap) look = 64u;
    // ... and room to stage ON when the last line runs past the look-ahead (fused_loop: a KiB at a time while the tile has room -- a line
    // that is not in the tile is parsed from global memory by one lane): a KiB, or another average line in a small tile
    uint64_t reserve = tile_cap >= 8192u ? 1024u : look;
    if (look + reserve + 1024u > tile_cap) reserve = 0u;
    const uint64_t room = tile_cap - kPreBytes - 16u - look - reserve;  // what the tile may hold besides
    // (less a sixteenth for a full wave of lines -- the count of sixty varies by one or two --, an eighth for a handful)
    uint64_t S = lines >= 32u ? (uint64_t)lines * avg_len * 15u / 16u : (uint64_t)lines * avg_len * 7u / 8u;
    if (S > room) S = room;
    S &= S >= 4096u ? ~255ull : ~15ull;
    if (S < 256u) S = room < 256u ? (room & ~15ull) : 256u;
    TilePlan p;
    p.S = (uint32_t)S;
    p.look = (uint32_t)look;
    return p;
}

}  // namespace fuse

// What a fused launch of one format looks like for lines of an average length (the stream's frames-per-byte experience of the ctx).
struct FusedGeom {
    uint32_t S = 0, look = 0;  // tile bytes, look-ahead bytes
    uint32_t tile = 0;         // LDS tile bytes (a multiple of 1024, >= 16 + S + look + 16)
    uint32_t ext = 1024;       // bytes staged on at a time when the tile's last line runs past the look-ahead (a multiple of 16, <= 1024)
    uint32_t L = 64;           // lines a pass of stage B takes
    uint32_t plan = 0;         // lines the tile is planned to hold (0: L -- one pass per tile); more than L: several passes per tile
    uint32_t variant = 0;      // the format's kernel instantiation (RFC5424: 1 = pair-parallel structured data; GELF: 1 = the constant 3 KiB geometry)
    bool ok = false;           // false: this stream keeps the separate framing pass (long lines: head staging)
};
inline FusedGeom fused_geometry(fg_format fmt, uint64_t avg_len, const fg_launch_opts& lo, bool link_bound) {
    FusedGeom g;
    if (avg_len < 1u) avg_len = 1u;
    const bool head = (lo.flags & FG_LO_FORCE_HEAD) || (avg_len >= 768u && !(lo.flags & FG_LO_NO_HEAD));
    uint32_t bound = 0;
    switch (fmt) {
        case FG_RFC5424:
            if (head) return g;
            g.variant = ((lo.flags & FG_LO_SD_PAIRS) || (avg_len >= 320u && !(lo.flags & FG_LO_SD_WALK))) ? 1u : 0u;
            g.L = 64u;
            // (the pair-parallel kernel's tile -- twice that across the link, where the grid is capped anyway and every byte of look-ahead
            //  is read twice over the link: the input side of the link is the ceiling there, 48 GB/s --; else the register window)
            bound = g.variant ? (link_bound ? 24576u : 12288u) : 16384u;
            break;
        case FG_LTSV:
            if (head) return g;
            g.L = 64u;
            bound = 18432u;
            break;
        case FG_GELF:
            // resident: eight lines to the pass (the row loop's geometry: more lines cost LDS, i.e. waves); across the link
            // the grid is capped anyway and the look-ahead is re-read over the link: thirty-two lines to the tile
            g.L = link_bound ? 48u : 8u;
            bound = link_bound ? 16384u : 8u * avg_len * 17u / 16u + 256u <= 3072u ? 3072u : 8192u;
            if (!link_bound && bound == 3072u && !lo.tile_cap && !lo.lines_per_group && !(lo.flags & FG_LO_GELF_GENERIC)) g.variant = 1u;
#if defined(FG_GELF_FUSED_PLAN)  // (A/B builds: resident tiles of that many lines, eight to the pass -- a quarter of the look-backs at 32.
                                 //  MEASURED, 4 M lines on one box: 5.14 ms as shipped, 7.46 with 16 lines, 7.18 with 32: profiles/r06at_gelf_fused_plan.log)
            if (!link_bound) {
                g.plan = FG_GELF_FUSED_PLAN;
                g.variant = 0u;
                bound = (uint32_t)((uint64_t)g.plan * avg_len + 2048u + 1023u) / 1024u * 1024u;
                if (bound > 16384u) bound = 16384u;
            }
#endif
            break;
        default:
            return g;
    }
    if (lo.lines_per_group >= 1u && lo.lines_per_group <= 64u) g.L = lo.lines_per_group;
    if (lo.tile_cap >= 1024u && lo.tile_cap <= 57344u) bound = (lo.tile_cap + 1023u) / 1024u * 1024u;
    if (bound < 2048u) bound = 2048u;
    if (link_bound) g.ext = 256u;  // (a KiB read on for a line that needed 150 bytes more is look-ahead too)
    if (lo.fused_ext >= 16u && lo.fused_ext <= 1024u) g.ext = lo.fused_ext & ~15u;
    const fuse::TilePlan tp = fuse::plan_tile(avg_len, g.plan ? g.plan : g.L, bound, lo.fused_look <= 4096u ? lo.fused_look : 0u);
    g.S = tp.S;
    g.look = tp.look;
    // (the tile keeps the room plan_tile left for staging on: the LDS tile is the bound it planned with unless the lines are short)
    g.tile = g.variant == 1u && fmt == FG_GELF ? 3072u : (fuse::kPreBytes + g.S + g.look + (bound >= 8192u ? 1024u : g.look) + 16u + 1023u) / 1024u * 1024u;
    if (g.tile > bound) g.tile = bound;
    if (g.tile < 4096u && !(g.variant == 1u && fmt == FG_GELF)) g.tile = 4096u;  // (plan_launch's floor)
    g.ok = g.S >= 16u && fuse::kPreBytes + g.S + g.look + 16u <= g.tile;
    return g;
}

// device scratch of one fused launch (zeroed by the launcher): ticket counters | total, abort | tile counts | block aggregates, prefixes
constexpr uint32_t kFusedCounters = 16;       // K ticket counters
constexpr uint32_t kFusedCounterStride = 32;  // u32 words between two of them (a 128-byte line each)
inline uint64_t fused_tiles(uint64_t nbytes, uint32_t S) { return (nbytes + S - 1u) / S; }
inline uint64_t fused_scratch_bytes(uint64_t nbytes, uint32_t S) {
    const uint64_t nt = fused_tiles(nbytes, S), nb = (nt + 63u) / 64u;
    return (uint64_t)kFusedCounters * kFusedCounterStride * 4u + 128u + ((nt * 4u + 63u) & ~63ull) + nb * 16u + 64u;
}

}  // namespace fg
