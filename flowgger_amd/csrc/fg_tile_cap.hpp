// fg_tile_cap.hpp -- the one piece of LAUNCH GEOMETRY the host side picks itself (the RFC3164 decoder's and the encoders' LDS tile; the
// streaming decoders plan theirs in fg_pipeline.hpp plan_launch).  Its own header so that a measured figure of those kernels
// (profiles/traffic.json, flowgger_amd/build.py source_hash) is keyed on this file and the kernel sources, not on the host pipelines.
// Included by fg_ctx.hpp (needs fg_ctx and up()).
#pragma once
namespace {
// LDS tile per 64-line wave: room for 64 average lines + 12.5 % + 512 B, 4..56 KiB (the kernel
// adds the space bitmap, 1/8 of the tile, on top).  fg_launch_opts::tile_cap overrides (bytes), for tuning.
uint32_t pick_tile_cap(const fg_ctx* ctx, uint64_t nbytes, uint64_t n, uint64_t max_cap, uint32_t margin_16ths = 2) {
    if (ctx->lo.tile_cap >= 1024 && ctx->lo.tile_cap <= max_cap) return (uint32_t)up(ctx->lo.tile_cap, 1024);
    uint64_t avg = n ? (nbytes + n - 1) / n : 0;
    uint64_t want = up(64 * avg * (16 + margin_16ths) / 16 + 512, 1024);
    if (want < 4096) want = 4096;
    if (want > max_cap) want = max_cap;
    return (uint32_t)want;
}

}  // namespace
