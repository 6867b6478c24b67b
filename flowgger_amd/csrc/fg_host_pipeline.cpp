// fg_host_pipeline.cpp -- the PCIe-inclusive HOST-BUFFER entry points of the C ABI (include/fg_hip.h): fg_decode_batch,
// fg_frame_decode_batch, fg_transcode_batch, the pinned allocator and fg_measure_link.  Everything here is copies, streams, events and
// calls of the device entry points (fg_capi.cpp); no kernel and no launch geometry lives in this file, so editing it does not change
// what a kernel does (flowgger_amd/build.py source_hash: measured HBM traffic is keyed on the kernels' own sources).
// Replaces the per-line `decoder.decode(line)` -> `encoder.encode(..)` -> `tx.send(..)` of handle_line for a whole batch
// (src/flowgger/splitter/line_splitter.rs:44-54).
#include <cstddef>

#include "fg_ctx.hpp"

extern "C" {

// ---- the batch buffers' allocator: page-locked memory through a process-wide POOL (a framer pinned max_bytes + 1 MiB per
// run(), i.e. per connection -- a thread-per-connection server with hundreds of clients pinned gigabytes and paid milliseconds of
// pin / unpin per connect).  A freed block is kept (up to `idle_limit` bytes in total) and handed to the next request it fits; the
// bytes pinned through this allocator are capped (`limit`): beyond the cap a request gets PAGEABLE memory -- fg_decode_batch and the
// raw-stream paths take either (pinned: zero-copy / link-speed uploads; pageable: the runtime's staged copies).
}  // extern "C"
#include <map>
#include <mutex>
namespace {
struct PinnedPool {
    std::mutex mu;
    std::map<void*, std::pair<uint64_t, bool>> live;  // block -> (capacity, pinned?)
    std::multimap<uint64_t, void*> idle;              // pinned blocks nobody holds, by capacity
    uint64_t pinned_bytes = 0, idle_bytes = 0;
    uint64_t limit = 8ull << 30, idle_limit = 256ull << 20;
};
PinnedPool& pinned_pool() {
    static PinnedPool* p = new PinnedPool();  // (never destroyed: no HIP calls from static destructors at process exit)
    return *p;
}
}  // namespace
extern "C" {

int fg_alloc_pinned(uint64_t bytes, void** out) {
    if (!out) return FG_ERR_ARG;
    *out = nullptr;
    const uint64_t need = up(bytes ? bytes : 1, 64u << 10);
    PinnedPool& P = pinned_pool();
    std::vector<void*> evict;  // idle blocks to unpin, once the lock is released
    bool pageable = false;
    {
        std::lock_guard<std::mutex> g(P.mu);
        auto it = P.idle.lower_bound(need);
        if (it != P.idle.end() && it->first <= 2u * need + (1u << 20)) {  // (a block of at most twice the size: no 64 MiB block for a 64 KiB request)
            void* p = it->second;
            const uint64_t cap = it->first;
            P.idle.erase(it);
            P.idle_bytes -= cap;
            P.live[p] = {cap, true};
            *out = p;
            return FG_OK;
        }
        if (P.pinned_bytes + need > P.limit) {
            // over the cap: what is idle goes first (unpinned BEHIND the lock: hipHostFree is a millisecond-scale call, and every connection
            // thread's alloc / free waits on this mutex -- ADVICE r5), then pageable memory
            while (!P.idle.empty() && P.pinned_bytes + need > P.limit) {
                auto last = std::prev(P.idle.end());
                evict.push_back(last->second);
                P.pinned_bytes -= last->first;
                P.idle_bytes -= last->first;
                P.idle.erase(last);
            }
            if (P.pinned_bytes + need > P.limit) {
                void* p = aligned_alloc(4096, (size_t)need);
                if (p) {
                    P.live[p] = {need, false};
                    *out = p;
                }
                pageable = true;
            }
        }
        if (!pageable) P.pinned_bytes += need;  // (reserved before the slow call, outside the lock)
    }
    for (void* e : evict) (void)hipHostFree(e);
    if (pageable) return *out ? FG_OK : FG_ERR_NOMEM;
    void* p = nullptr;
    if (hipHostMalloc(&p, need, hipHostMallocDefault) != hipSuccess || !p) {
        (void)hipGetLastError();
        std::lock_guard<std::mutex> g(P.mu);
        P.pinned_bytes -= need;
        return FG_ERR_HIP;
    }
    std::lock_guard<std::mutex> g(P.mu);
    P.live[p] = {need, true};
    *out = p;
    return FG_OK;
}
void fg_free_pinned(void* p) {
    if (!p) return;
    PinnedPool& P = pinned_pool();
    bool unpin = false;
    {
        std::lock_guard<std::mutex> g(P.mu);
        auto it = P.live.find(p);
        if (it == P.live.end()) return;  // (not ours)
        const uint64_t cap = it->second.first;
        const bool pinned = it->second.second;
        P.live.erase(it);
        if (!pinned) {
            free(p);
            return;
        }
        if (P.idle_bytes + cap <= P.idle_limit) {
            P.idle.emplace(cap, p);
            P.idle_bytes += cap;
        } else {
            P.pinned_bytes -= cap;
            unpin = true;
        }
    }
    if (unpin) (void)hipHostFree(p);
}
int fg_set_pinned_limits(uint64_t total_bytes, uint64_t idle_bytes) {
    PinnedPool& P = pinned_pool();
    std::vector<void*> evict;
    {
        std::lock_guard<std::mutex> g(P.mu);
        P.limit = total_bytes;
        P.idle_limit = idle_bytes;
        while (!P.idle.empty() && P.idle_bytes > P.idle_limit) {
            auto last = std::prev(P.idle.end());
            evict.push_back(last->second);
            P.pinned_bytes -= last->first;
            P.idle_bytes -= last->first;
            P.idle.erase(last);
        }
    }
    for (void* e : evict) (void)hipHostFree(e);  // (behind the lock)
    return FG_OK;
}
// 1: p is a block of this allocator that is page-locked; 0: one that fell back to pageable memory (the cap was reached: zero-copy and
// link-speed uploads do not apply to it); -1: not a block of this allocator
int fg_is_pinned(const void* p) {
    PinnedPool& P = pinned_pool();
    std::lock_guard<std::mutex> g(P.mu);
    auto it = P.live.find(const_cast<void*>(p));
    return it == P.live.end() ? -1 : it->second.second ? 1 : 0;
}
int fg_pinned_stats(uint64_t* pinned_bytes, uint64_t* idle_bytes, uint64_t* live_blocks) {
    PinnedPool& P = pinned_pool();
    std::lock_guard<std::mutex> g(P.mu);
    if (pinned_bytes) *pinned_bytes = P.pinned_bytes;
    if (idle_bytes) *idle_bytes = P.idle_bytes;
    if (live_blocks) *live_blocks = P.live.size();
    return FG_OK;
}

// Streams and events of the pipelined host paths (created on first use).  Three roles, each its own stream, chained by events per slice:
//   s_run  = the ctx's OWN stream (the first one it created): kernels, and the 8-byte counter copies that follow them
//   s_up   = stream2: every host -> device copy, back to back, nothing else ever queued between two of them
//   s_down = stream3: every device -> host copy of table columns
// so that the link carries both directions at once.  (Round 2 alternated whole slices -- H2D, kernel, D2H -- between two streams:
// measured, that gave the rate of NO overlap at all, 42 of 57 GB/s.)  One platform observation decides an ORDER below (MI355X /
// ROCm 7.2, tools/probe/stream_pairs.cpp and four orderings of this code): the runtime binds its copy paths to streams as they first
// copy, and with stream2 / stream3 as the first streams of the process to copy, fg_decode_batch ran at 160 M lines/s; with the ctx's
// own stream having copied a few bytes each way BEFORE them, at 198 M (the figure DESIGN.md quotes belongs to this ordering).  Hence
// the probe copy on ctx->stream.  fg_measure_link times its copies on s_up / s_down as assigned here.
static int ensure_pipeline(fg_ctx* ctx, uint32_t slices) {
    if (!ctx->stream2) FG_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
    if (!ctx->ev_ready) FG_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_ready, hipEventDisableTiming));  // (its own check: a failed create is retried)
    if (!ctx->stream3) FG_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream3, hipStreamNonBlocking));
    while (ctx->ev_slice.size() < 2ull * slices) {
        hipEvent_t e = nullptr;
        FG_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->ev_slice.push_back(e);
    }
    if (!ctx->s_up) {
        // (see above: the ctx's own stream copies a few bytes each way before stream2 / stream3 are ever used)
        uint64_t probe = 0;
        uint64_t* d_probe = nullptr;
        FG_HIP(ctx, hipMalloc((void**)&d_probe, 8));
        (void)hipMemcpyAsync(d_probe, &probe, 8, hipMemcpyHostToDevice, ctx->stream);
        (void)hipMemcpyAsync(&probe, d_probe, 8, hipMemcpyDeviceToHost, ctx->stream);
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(d_probe);
    }
    ctx->s_up = ctx->stream2;
    ctx->s_run = ctx->stream;
    ctx->s_down = ctx->stream3;
    return FG_OK;
}
// slices of a host batch: small enough that filling and draining the pipeline costs little (1/8 of the batch at most), large
// enough that a slice's fourteen API calls stay far below its transfer time
static uint32_t slice_count(uint64_t nbytes, uint64_t n) {
    if (nbytes < (32ull << 20)) return 1;  // small batches: one stream, no events (a batch of one line costs what it did)
    uint64_t slice = nbytes / 8;
    if (slice < (8ull << 20)) slice = 8ull << 20;
    if (slice > (32ull << 20)) slice = 32ull << 20;
    uint64_t k = (nbytes + slice - 1) / slice;
    if (k < 1) k = 1;
    if (k > 256) k = 256;
    if (n < k) k = n ? n : 1;
    return (uint32_t)k;
}

int fg_measure_link(fg_ctx* ctx, uint64_t nbytes, double gbps[3]) {
    if (!ctx || !gbps || nbytes < 4096) return FG_ERR_ARG;
    DeviceGuard g(ctx->device);
    {   // on the very streams the pipelined host paths copy on: uploads on s_up (stream2), downloads on s_down (stream3)
        const int prc = ensure_pipeline(ctx, 1);
        if (prc != FG_OK) return prc;
    }
    const hipStream_t s_up = ctx->s_up, s_down = ctx->s_down;
    uint8_t *h0 = nullptr, *h1 = nullptr, *d0 = nullptr, *d1 = nullptr;
    hipEvent_t e[4] = {nullptr, nullptr, nullptr, nullptr};
    int rc = FG_OK;
    auto fail = [&](hipError_t err) {
        if (err != hipSuccess && rc == FG_OK) {
            ctx->last_hip = (int)err;
            rc = FG_ERR_HIP;
        }
        return err != hipSuccess;
    };
    do {
        if (fail(hipHostMalloc((void**)&h0, nbytes, hipHostMallocDefault)) || fail(hipHostMalloc((void**)&h1, nbytes, hipHostMallocDefault))) break;
        if (fail(hipMalloc((void**)&d0, nbytes)) || fail(hipMalloc((void**)&d1, nbytes))) break;
        memset(h0, 0x5A, nbytes);  // (touch the pages: first use must not be part of the figure)
        memset(h1, 0, nbytes);
        bool bad = false;
        for (auto& ev : e) bad = bad || fail(hipEventCreate(&ev));
        if (bad) break;
        double best[3] = {0, 0, 0};
        for (int rep = 0; rep < 4 && rc == FG_OK; ++rep) {  // (rep 0 warms the path up)
            float ms = 0.f;
            // host -> device
            if (fail(hipEventRecord(e[0], s_up)) || fail(hipMemcpyAsync(d0, h0, nbytes, hipMemcpyHostToDevice, s_up)) ||
                fail(hipEventRecord(e[1], s_up)) || fail(hipEventSynchronize(e[1])) || fail(hipEventElapsedTime(&ms, e[0], e[1])))
                break;
            if (rep && ms > 0.f) best[0] = std::max(best[0], (double)nbytes / (ms * 1e-3) / 1e9);
            // device -> host
            if (fail(hipEventRecord(e[0], s_down)) || fail(hipMemcpyAsync(h1, d1, nbytes, hipMemcpyDeviceToHost, s_down)) ||
                fail(hipEventRecord(e[1], s_down)) || fail(hipEventSynchronize(e[1])) || fail(hipEventElapsedTime(&ms, e[0], e[1])))
                break;
            if (rep && ms > 0.f) best[1] = std::max(best[1], (double)nbytes / (ms * 1e-3) / 1e9);
            // both at once: the slower stream bounds the pair (wall clock around both)
            if (fail(hipStreamSynchronize(s_up)) || fail(hipStreamSynchronize(s_down))) break;
            timespec a, b;
            clock_gettime(CLOCK_MONOTONIC, &a);
            if (fail(hipMemcpyAsync(d0, h0, nbytes, hipMemcpyHostToDevice, s_up)) ||
                fail(hipMemcpyAsync(h1, d1, nbytes, hipMemcpyDeviceToHost, s_down)) || fail(hipStreamSynchronize(s_up)) ||
                fail(hipStreamSynchronize(s_down)))
                break;
            clock_gettime(CLOCK_MONOTONIC, &b);
            const double s = (double)(b.tv_sec - a.tv_sec) + (double)(b.tv_nsec - a.tv_nsec) * 1e-9;
            if (rep && s > 0) best[2] = std::max(best[2], 2.0 * (double)nbytes / s / 1e9);
        }
        for (int k = 0; k < 3; ++k) gbps[k] = best[k];
    } while (false);
    for (auto& ev : e)
        if (ev) (void)hipEventDestroy(ev);
    if (d0) (void)hipFree(d0);
    if (d1) (void)hipFree(d1);
    if (h0) (void)hipHostFree(h0);
    if (h1) (void)hipHostFree(h1);
    return rc;
}

// Is [p, p + n) pinned host memory that the device can address (hipHostMalloc / hipHostRegister)?  -> the device's view of p, else null.
static const void* device_view_of_pinned(const void* p) {
    if (!p) return nullptr;
    hipPointerAttribute_t a;
    memset(&a, 0, sizeof a);
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();  // (pageable memory: the runtime reports an error and keeps it sticky)
        return nullptr;
    }
    return a.type == hipMemoryTypeHost ? a.devicePointer : nullptr;
}

static int pinned_tables_for_kernels(fg_ctx* ctx, const fg_tables& ht, fg_tables* kt);

// ZERO-COPY form of fg_decode_batch: when the caller's bytes and offsets live in PINNED host memory (where a batching framer
// accumulates them: fg_alloc_pinned) the decode kernels read them in place over the link and write every table column straight
// into the ctx's pinned tables -- no hipMemcpy, no slices, no events: ONE launch.  The kernels stream 1 KiB per wave-instruction
// with a 20 KiB prefetch window per wave, far more in flight than the link's latency-bandwidth product needs, so the launch runs at
// link speed in BOTH directions at once (tools/probe/zero_copy.py, profiles/r04b_zero_copy_probe.json: GELF 153 vs 134 M lines/s,
// LTSV 194 vs 170, cfg2 210 vs 196 against the sliced hipMemcpy pipeline below, whose uploads and downloads the runtime of this
// platform queues on ONE copy engine: profiles/r04a_timeline_*.log, an upload and a download never in flight together).  Only the
// entry counter stays in HBM (atomics).  Returns FG_ERR_UNSUPPORTED when the buffers do not qualify: the caller falls through.
// A decode grid whose tables lie in pinned host memory runs at the LINK's pace: its waves mostly wait for their stores to drain.  At full
// occupancy (GELF: twenty waves per CU) that costs throughput -- same box, 4 M GELF lines: fg_frame_decode_batch 105 M lines/s with
// every wave slot taken, 129 M at eight waves per CU; fg_decode_batch 151 -> 161 M (profiles/r04w_frame_overlap.log; the other formats'
// kernels hold eight waves or fewer anyway).  For the duration of a host pipeline the grid is capped at eight, unless the caller set
// its own figure (fg_set_launch_opts).
// the launches of a sliced batch share ONE entry table: fg::entry_chunk gives each of them a 1 / shares part of the table's budget
struct TableShares {
    fg_ctx* ctx;
    TableShares(fg_ctx* c, uint32_t launches) : ctx(c) { ctx->table_shares = launches; }
    ~TableShares() { ctx->table_shares = 0; }
};
struct LinkBoundGrid {  // (a per-call cap beside the caller's launch options, never written into them: ADVICE r4)
    fg_ctx* ctx;
    explicit LinkBoundGrid(fg_ctx* c, uint32_t waves = 8) : ctx(c) { ctx->link_bound_waves = waves; }
    ~LinkBoundGrid() { ctx->link_bound_waves = 0; }
};

static int decode_batch_zero_copy(fg_ctx* ctx, fg_format fmt, const uint8_t* bytes, uint64_t nbytes, const uint64_t* offsets, uint64_t n,
                                  fg_tables* out) {
    if (n == 0 || (ctx->lo.flags & FG_LO_NO_ZERO_COPY)) return FG_ERR_UNSUPPORTED;
    // (reads AND writes cross the link: two waves per CU fill it, and every wave more only queues in front of the dependent reads --
    //  offsets before bytes, GELF's exact form: LTSV 0.81 -> 0.89 of the link, structured data 0.90 -> 0.93, GELF 0.85 -> 0.87 with four,
    //  profiles/r06aa_decode_batch_waves.log)
    LinkBoundGrid grid(ctx, fmt == FG_GELF ? 4u : 2u);
    const uint8_t* d_bytes = (const uint8_t*)device_view_of_pinned(bytes);
    const uint64_t* d_offsets = (const uint64_t*)device_view_of_pinned(offsets);
    if (!d_bytes || !d_offsets || ((uintptr_t)d_bytes & 15u) != 0) return FG_ERR_UNSUPPORTED;
    {
        // ONE mapping with a constant device-view delta from the first byte to the last the kernels touch (the first and
        // the last byte of two adjacent registrations both qualify, the bytes between their device views need not be contiguous).
        // The kernels' 16-byte loads reach up to nbytes rounded up to 16 (include/fg_hip.h: the packed buffer must be readable that far):
        // that byte must be part of the same mapping too.
        const uint8_t* last_b = bytes + ((nbytes + 15u) & ~15ull) - 1u;
        const uint8_t* dv_last_b = (const uint8_t*)device_view_of_pinned(last_b);
        const uint8_t* dv_last_o = (const uint8_t*)device_view_of_pinned(offsets + n);
        if (!dv_last_b || !dv_last_o || dv_last_b - d_bytes != last_b - bytes ||
            dv_last_o - (const uint8_t*)d_offsets != (const uint8_t*)(offsets + n) - (const uint8_t*)offsets)
            return FG_ERR_UNSUPPORTED;
    }
    int rc;
    uint64_t ent_cap = fmt == FG_RFC5424 ? nbytes / 16 + 1024 : nbytes / 8 + 1024;
    if (fmt == FG_RFC3164) ent_cap = 16;  // RFC3164 produces no entries
    const hipStream_t s = ctx->stream;
    for (;;) {
        if (ent_cap > 0xFFFFFFF0ull) ent_cap = 0xFFFFFFF0ull;
        uint64_t total = 0;
        carve(nullptr, n, ent_cap, nullptr, &total);
        if ((rc = grow_pinned(ctx, (void**)&ctx->h_tab, &ctx->h_tab_cap, total)) != FG_OK) return rc;
        fg_tables ht;
        carve(ctx->h_tab, n, ent_cap, &ht, nullptr);
        fg_tables kt;  // what the kernels see: the pinned columns (their device view) and the counter in HBM
        if ((rc = pinned_tables_for_kernels(ctx, ht, &kt)) != FG_OK) return rc;
        if ((rc = fg_decode_frames_impl(ctx, fmt, FG_FRAME_NONE, d_bytes, nbytes, d_offsets, n, nullptr, &kt, (void*)s, true,
                                        offsets[n] - offsets[0])) != FG_OK)
            return rc;
        uint64_t used = 0;
        FG_HIP(ctx, hipMemcpyAsync(&used, ctx->d_used, 8, hipMemcpyDeviceToHost, s));
        FG_HIP(ctx, hipStreamSynchronize(s));
        if (used > ent_cap) {
            if (ent_cap >= 0xFFFFFFF0ull) return FG_ERR_ENT_OVERFLOW;
            ent_cap = used + used / 8 + 1024;
            continue;
        }
        *ht.ent_used = used;
        *out = ht;
        ctx->last_host_path = FG_PATH_DECODE_ZERO_COPY;
        return FG_OK;
    }
}

int fg_decode_batch(fg_ctx* ctx, fg_format fmt, const uint8_t* bytes, uint64_t nbytes, const uint64_t* offsets,
                    uint64_t n, fg_tables* out) {
    if (!ctx || !out || (n && !offsets) || (nbytes && !bytes)) return FG_ERR_ARG;
    if (n && (offsets[n] > nbytes || offsets[0] > offsets[n])) return FG_ERR_ARG;
    DeviceGuard g(ctx->device);
    int rc;
    if ((rc = decode_batch_zero_copy(ctx, fmt, bytes, nbytes, offsets, n, out)) != FG_ERR_UNSUPPORTED) return rc;
    const uint32_t slices = slice_count(nbytes, n);
    TableShares shares(ctx, slices);
    if (slices > 1 && (rc = ensure_pipeline(ctx, slices)) != FG_OK) return rc;  // (a small batch stays on the ctx's own stream)
    if (slices > 1 && !ctx->h_cnt) FG_HIP(ctx, hipHostMalloc((void**)&ctx->h_cnt, 65536, hipHostMallocDefault));
    if ((rc = grow_dev(ctx, (void**)&ctx->d_bytes, &ctx->d_bytes_cap, up(nbytes, 16) + 16)) != FG_OK) return rc;
    if ((rc = grow_dev(ctx, (void**)&ctx->d_offsets, &ctx->d_offsets_cap, (n + 1) * 8)) != FG_OK) return rc;
    std::vector<uint64_t> cut(slices + 1);
    if (fg_shard_plan(offsets, n, slices, cut.data()) != FG_OK) return FG_ERR_ARG;
    const bool piped = slices > 1;
    const hipStream_t s_down = piped ? ctx->s_down : ctx->stream, s_up = piped ? ctx->s_up : s_down, s_run = piped ? ctx->s_run : s_down;
    auto drain = [&]() {
        (void)hipStreamSynchronize(s_up);
        (void)hipStreamSynchronize(s_run);
        (void)hipStreamSynchronize(s_down);
    };
    // entry capacity: start from one entry per 16 (RFC5424) / 8 input bytes, grow on overflow
    uint64_t ent_cap = fmt == FG_RFC5424 ? nbytes / 16 + 1024 : nbytes / 8 + 1024;
    if (fmt == FG_RFC3164) ent_cap = 16;  // RFC3164 produces no entries
    for (;;) {
        if (ent_cap > 0xFFFFFFF0ull) ent_cap = 0xFFFFFFF0ull;
        uint64_t total = 0;
        carve(nullptr, n, ent_cap, nullptr, &total);
        if ((rc = grow_dev(ctx, (void**)&ctx->d_tab, &ctx->d_tab_cap, total)) != FG_OK) return rc;
        if ((rc = grow_pinned(ctx, (void**)&ctx->h_tab, &ctx->h_tab_cap, total)) != FG_OK) return rc;
        fg_tables dt, ht;
        carve(ctx->d_tab, n, ent_cap, &dt, nullptr);
        carve(ctx->h_tab, n, ent_cap, &ht, nullptr);
        FG_HIP(ctx, hipMemsetAsync(dt.ent_used, 0, 8, s_run));
        // Rows land at their final position, entries share one counter: the result is the same as one monolithic launch.
        // A slice is ISSUED (upload, kernels, the entry counter's value after it into a pinned word) and later COLLECTED (its rows and
        // the entries its kernels appended -- the range between two counter values -- come back on the download stream).  The host
        // issues sixteen slices ahead of the one it collects, so the link has uploads queued at all times and every table, the
        // entry columns included, crosses it while later slices are still going up.
        uint64_t* const ent_cnt = piped ? ctx->h_cnt + 1024 : nullptr;
        bool first_issued = false;  // offsets[l0] of a slice belongs to the slice before it -- except for the first slice that has rows
        fg_span* hs[6] = {ht.hostname, ht.appname, ht.procid, ht.msgid, ht.msg, ht.full_msg};
        fg_span* ds[6] = {dt.hostname, dt.appname, dt.procid, dt.msgid, dt.msg, dt.full_msg};
        auto issue = [&](uint32_t k) -> int {
            const uint64_t l0 = cut[k], l1 = cut[k + 1], rows = l1 - l0;
            if (rows == 0) return FG_OK;
            // ---- upload: the slice's offsets (the first slice WITH ROWS also takes its own start: fg_shard_plan gives empty leading
            //      slices when the lines cover fewer bytes than there are slices -- ADVICE r3) and its bytes, copied on 16-byte
            //      boundaries (the neighbouring bytes are the same data)
            const uint64_t o0 = first_issued ? l0 + 1 : l0;
            first_issued = true;
            FG_HIP(ctx, hipMemcpyAsync(ctx->d_offsets + o0, offsets + o0, (l1 + 1 - o0) * 8, hipMemcpyHostToDevice, s_up));
            const uint64_t b0 = offsets[l0] & ~15ull, b1 = offsets[l1];
            if (b1 > b0) FG_HIP(ctx, hipMemcpyAsync(ctx->d_bytes + b0, bytes + b0, b1 - b0, hipMemcpyHostToDevice, s_up));
            if (piped) {
                FG_HIP(ctx, hipEventRecord(ctx->ev_slice[2 * k], s_up));
                FG_HIP(ctx, hipStreamWaitEvent(s_run, ctx->ev_slice[2 * k], 0));
            }
            // ---- decode
            fg_tables sl = dt;  // the slice's rows: same arrays, shifted by l0
            sl.n = rows;
            sl.meta += l0;
            sl.ts += l0;
            sl.hostname += l0;
            sl.appname += l0;
            sl.procid += l0;
            sl.msgid += l0;
            sl.msg += l0;
            sl.full_msg += l0;
            sl.ent_first += l0;
            sl.ent_count += l0;
            const int drc = fg_decode_frames_impl(ctx, fmt, FG_FRAME_NONE, ctx->d_bytes, nbytes, ctx->d_offsets + l0, rows, nullptr, &sl, (void*)s_run, false,
                                               offsets[l1] - offsets[l0]);
            if (drc != FG_OK) return drc;
            if (piped) {
                FG_HIP(ctx, hipMemcpyAsync(ent_cnt + k, dt.ent_used, 8, hipMemcpyDeviceToHost, s_run));
                FG_HIP(ctx, hipEventRecord(ctx->ev_slice[2 * k + 1], s_run));
            }
            return FG_OK;
        };
        auto download_rows = [&](uint64_t l0, uint64_t rows) -> int {
            FG_HIP(ctx, hipMemcpyAsync(ht.meta + l0, dt.meta + l0, rows * 4, hipMemcpyDeviceToHost, s_down));
            FG_HIP(ctx, hipMemcpyAsync(ht.ts + l0, dt.ts + l0, rows * 8, hipMemcpyDeviceToHost, s_down));
            for (int j = 0; j < 6; ++j) FG_HIP(ctx, hipMemcpyAsync(hs[j] + l0, ds[j] + l0, rows * 8, hipMemcpyDeviceToHost, s_down));
            FG_HIP(ctx, hipMemcpyAsync(ht.ent_first + l0, dt.ent_first + l0, rows * 4, hipMemcpyDeviceToHost, s_down));
            FG_HIP(ctx, hipMemcpyAsync(ht.ent_count + l0, dt.ent_count + l0, rows * 4, hipMemcpyDeviceToHost, s_down));
            return FG_OK;
        };
        auto download_entries = [&](uint64_t e0, uint64_t e1) -> int {
            if (e1 <= e0) return FG_OK;
            FG_HIP(ctx, hipMemcpyAsync(ht.ent_name + e0, dt.ent_name + e0, (e1 - e0) * 8, hipMemcpyDeviceToHost, s_down));
            FG_HIP(ctx, hipMemcpyAsync(ht.ent_val + e0, dt.ent_val + e0, (e1 - e0) * 8, hipMemcpyDeviceToHost, s_down));
            FG_HIP(ctx, hipMemcpyAsync(ht.ent_type + e0, dt.ent_type + e0, e1 - e0, hipMemcpyDeviceToHost, s_down));
            FG_HIP(ctx, hipMemcpyAsync(ht.ent_flags + e0, dt.ent_flags + e0, e1 - e0, hipMemcpyDeviceToHost, s_down));
            return FG_OK;
        };
        uint64_t used = 0;
        bool overflow = false;
        uint32_t issued = 0;
        for (uint32_t k = 0; k < slices && n; ++k) {
            while (issued < slices && issued < k + 16) {
                if ((rc = issue(issued)) != FG_OK) {  // copies of earlier slices may still be in flight into h_tab / d_tab
                    drain();
                    return rc;
                }
                ++issued;
            }
            const uint64_t l0 = cut[k], rows = cut[k + 1] - l0;
            if (rows == 0) continue;
            if (piped) {
                if (hipEventSynchronize(ctx->ev_slice[2 * k + 1]) != hipSuccess) {
                    drain();
                    return FG_ERR_HIP;
                }
                const uint64_t cnt = ent_cnt[k];  // the counter after this slice's kernels: its entries are [used, cnt)
                if (cnt > ent_cap) {
                    overflow = true;
                    break;
                }
                if ((rc = download_rows(l0, rows)) != FG_OK || (rc = download_entries(used, cnt)) != FG_OK) {
                    drain();
                    return rc;
                }
                used = cnt;
            } else if ((rc = download_rows(l0, rows)) != FG_OK) {
                drain();
                return rc;
            }
        }
        if (!piped || overflow) {  // one stream (a small batch), or the entry table ran out: the counter's final value
            while (overflow && issued < slices) {  // (kernels past the capacity still count: the counter then says what the batch needs)
                if ((rc = issue(issued)) != FG_OK) {
                    drain();
                    return rc;
                }
                ++issued;
            }
            if (overflow) drain();
            FG_HIP(ctx, hipMemcpyAsync(&used, dt.ent_used, 8, hipMemcpyDeviceToHost, s_run));
            FG_HIP(ctx, hipStreamSynchronize(s_run));
            overflow = used > ent_cap;
        }
        if (overflow) {
            drain();
            if (ent_cap >= 0xFFFFFFF0ull) return FG_ERR_ENT_OVERFLOW;
            ent_cap = used + used / 8 + 1024;
            continue;
        }
        if (!piped && (rc = download_entries(0, used)) != FG_OK) {
            drain();
            return rc;
        }
        *ht.ent_used = used;
        FG_HIP(ctx, hipStreamSynchronize(s_down));
        FG_HIP(ctx, hipStreamSynchronize(s_run));
        *out = ht;
        ctx->last_host_path = FG_PATH_DECODE_SLICED;
        return FG_OK;
    }
}

static int frame_stage(fg_ctx* ctx, fg_framing framing, uint64_t nbytes, int final, uint64_t* n_frames, uint64_t* consumed);

static int frame_decode_one_piece(fg_ctx* ctx, fg_format fmt, fg_framing framing, const uint8_t* bytes, uint64_t nbytes, int final,
                                  fg_tables* out, const uint64_t** out_offsets, uint64_t* n_frames, uint64_t* consumed) {
    if (!ctx || !out || !out_offsets || !n_frames || !consumed || (nbytes && !bytes)) return FG_ERR_ARG;
    if (framing != FG_FRAME_LINE && framing != FG_FRAME_NUL) return FG_ERR_UNSUPPORTED;
    *n_frames = 0;
    *consumed = 0;
    *out_offsets = nullptr;
    if (nbytes == 0) return FG_OK;
    DeviceGuard g(ctx->device);
    hipStream_t s = ctx->stream;
    int rc;
    if ((rc = grow_dev(ctx, (void**)&ctx->d_bytes, &ctx->d_bytes_cap, up(nbytes, 16) + 16)) != FG_OK) return rc;
    FG_HIP(ctx, hipMemcpyAsync(ctx->d_bytes, bytes, nbytes, hipMemcpyHostToDevice, s));
    FG_HIP(ctx, hipMemsetAsync(ctx->d_bytes + nbytes, 0, up(nbytes, 16) + 16 - nbytes, s));
    // 1. frame: offsets + UTF-8 verdicts
    uint64_t n = 0;
    if ((rc = frame_stage(ctx, framing, nbytes, final, &n, consumed)) != FG_OK) return rc;
    *n_frames = n;
    if (n && nbytes >= (1u << 20)) ctx->frames_per_byte = (double)n / (double)nbytes;
    if ((n + 1) * 8 > ctx->h_off_cap) {
        if (ctx->h_off) FG_HIP(ctx, hipHostFree(ctx->h_off));
        ctx->h_off = nullptr;
        ctx->h_off_cap = 0;
        uint64_t want = up((n + 1) * 8 + (n + 1) * 2, 1 << 16);
        FG_HIP(ctx, hipHostMalloc((void**)&ctx->h_off, want, hipHostMallocDefault));
        ctx->h_off_cap = want;
    }
    FG_HIP(ctx, hipMemcpyAsync(ctx->h_off, ctx->d_offsets, (n + 1) * 8, hipMemcpyDeviceToHost, s));
    *out_offsets = ctx->h_off;
    if (n == 0) {
        FG_HIP(ctx, hipStreamSynchronize(s));
        fg_tables empty{};
        *out = empty;
        return FG_OK;
    }
    // 2. decode the frames in place (terminators stripped in-kernel, invalid UTF-8 -> FG_ST_BAD_UTF8)
    const uint64_t used_bytes = *consumed;
    uint64_t ent_cap = fmt == FG_RFC5424 ? used_bytes / 16 + 1024 : used_bytes / 8 + 1024;
    for (;;) {
        if (ent_cap > 0xFFFFFFF0ull) ent_cap = 0xFFFFFFF0ull;
        uint64_t bytes_total = 0;
        carve(nullptr, n, ent_cap, nullptr, &bytes_total);
        if ((rc = grow_dev(ctx, (void**)&ctx->d_tab, &ctx->d_tab_cap, bytes_total)) != FG_OK) return rc;
        if (bytes_total > ctx->h_tab_cap) {
            if (ctx->h_tab) FG_HIP(ctx, hipHostFree(ctx->h_tab));
            ctx->h_tab = nullptr;
            ctx->h_tab_cap = 0;
            uint64_t want = up(bytes_total + bytes_total / 4, 1 << 20);
            FG_HIP(ctx, hipHostMalloc((void**)&ctx->h_tab, want, hipHostMallocDefault));
            ctx->h_tab_cap = want;
        }
        fg_tables dt, ht;
        carve(ctx->d_tab, n, ent_cap, &dt, nullptr);
        carve(ctx->h_tab, n, ent_cap, &ht, nullptr);
        rc = fg_decode_frames_device(ctx, fmt, framing, ctx->d_bytes, used_bytes, ctx->d_offsets, n, ctx->d_bad, &dt, FG_STREAM_OWN);
        if (rc != FG_OK) return rc;
        uint64_t used = 0;
        FG_HIP(ctx, hipMemcpyAsync(&used, dt.ent_used, 8, hipMemcpyDeviceToHost, s));
        FG_HIP(ctx, hipStreamSynchronize(s));
        if (used > ent_cap) {
            if (ent_cap >= 0xFFFFFFF0ull) return FG_ERR_ENT_OVERFLOW;
            ent_cap = used + used / 8 + 1024;
            continue;
        }
        uint64_t sizes[FG_TABLE_ARRAYS];
        fg_tables_layout(n, used, sizes);
        void* dsts[FG_TABLE_ARRAYS] = {ht.meta, ht.ts, ht.hostname, ht.appname, ht.procid, ht.msgid, ht.msg, ht.full_msg,
                                       ht.ent_first, ht.ent_count, ht.ent_name, ht.ent_val, ht.ent_type, ht.ent_flags, ht.ent_used};
        void* srcs[FG_TABLE_ARRAYS] = {dt.meta, dt.ts, dt.hostname, dt.appname, dt.procid, dt.msgid, dt.msg, dt.full_msg,
                                       dt.ent_first, dt.ent_count, dt.ent_name, dt.ent_val, dt.ent_type, dt.ent_flags, dt.ent_used};
        for (int k = 0; k < FG_TABLE_ARRAYS; ++k)
            if (sizes[k]) FG_HIP(ctx, hipMemcpyAsync(dsts[k], srcs[k], sizes[k], hipMemcpyDeviceToHost, s));
        FG_HIP(ctx, hipStreamSynchronize(s));
        *out = ht;
        return FG_OK;
    }
}

// the kernels' view of tables carved from the ctx's PINNED host block (zero-copy output): every column pointer translated to the
// device's view of that memory, the entry counter in HBM (atomics)
static int pinned_tables_for_kernels(fg_ctx* ctx, const fg_tables& ht, fg_tables* kt) {
    const uint8_t* dv = (const uint8_t*)device_view_of_pinned(ctx->h_tab);
    if (!dv) return FG_ERR_UNSUPPORTED;
    if (!ctx->d_used) FG_HIP(ctx, hipMalloc((void**)&ctx->d_used, 256));
    const ptrdiff_t delta = dv - ctx->h_tab;  // (0 under unified addressing)
    *kt = ht;
    kt->meta = (uint32_t*)((uint8_t*)ht.meta + delta);
    kt->ts = (double*)((uint8_t*)ht.ts + delta);
    kt->hostname = (fg_span*)((uint8_t*)ht.hostname + delta);
    kt->appname = (fg_span*)((uint8_t*)ht.appname + delta);
    kt->procid = (fg_span*)((uint8_t*)ht.procid + delta);
    kt->msgid = (fg_span*)((uint8_t*)ht.msgid + delta);
    kt->msg = (fg_span*)((uint8_t*)ht.msg + delta);
    kt->full_msg = (fg_span*)((uint8_t*)ht.full_msg + delta);
    kt->ent_first = (uint32_t*)((uint8_t*)ht.ent_first + delta);
    kt->ent_count = (uint32_t*)((uint8_t*)ht.ent_count + delta);
    kt->ent_name = (fg_span*)((uint8_t*)ht.ent_name + delta);
    kt->ent_val = (uint64_t*)((uint8_t*)ht.ent_val + delta);
    kt->ent_type = (uint8_t*)ht.ent_type + delta;
    kt->ent_flags = (uint8_t*)ht.ent_flags + delta;
    kt->ent_used = ctx->d_used;
    return FG_OK;
}

// fg_frame_decode_batch for a LARGE raw chunk: the chunk crosses the link in slices (multiples of the framing kernels' 16 KiB
// block) on the upload stream; as soon as a slice is there it is framed (delimiter ranks continue where the slice before stopped),
// its frame count comes back to the host through a pinned word, and the frames that END in it are decoded -- all while the next
// slices are still on the link.  The host never frames, never uploads offsets.  Round 4: the decode kernels write the table
// columns STRAIGHT INTO THE PINNED HOST TABLES (zero-copy output) and the per-slice counters come back by a one-thread kernel, so
// the copy engine carries nothing but the uploads -- on this platform a device -> host hipMemcpy queues behind every upload issued
// before it (profiles/r04a_timeline_*: an upload and a download were never in flight together), which is what held the GELF and
// LTSV corpora at the SUM of their two directions.  Only the frame offsets (8 bytes per frame) are still copied.
// Tables are sized from what the ctx's last chunk held; a chunk that outgrows the estimate (or the entry table) returns
// FG_ERR_UNSUPPORTED and takes the one-piece path, which counts first.
static int frame_decode_sliced(fg_ctx* ctx, fg_format fmt, fg_framing framing, const uint8_t* bytes, uint64_t nbytes, int final,
                               fg_tables* out, const uint64_t** out_offsets, uint64_t* n_frames, uint64_t* consumed) {
    int rc;
    LinkBoundGrid grid(ctx);
    // GELF's exact form (k_gelf_general: the lines the fast form hands back) runs ONCE, behind the last slice, over all rows -- it is a
    // chain of dependent steps of ~170 us however few lines it gets, and launched per slice it was a quarter of this path's GPU time
    struct DeferGeneral {
        fg_ctx* ctx;
        explicit DeferGeneral(fg_ctx* c, bool on) : ctx(c) {
            ctx->defer_general = on;
            ctx->batch_epoch = 0;
        }
        ~DeferGeneral() {  // (also on every error path: nothing of this batch is left pending in the ctx)
            ctx->defer_general = false;
            ctx->batch_epoch = 0;
        }
    } defer(ctx, fmt == FG_GELF);
    const uint64_t blk = fg_frame_block_bytes();
    // (sixteen slices or more: the first upload and the last decode + download run alone -- a sixteenth of the batch each, not an eighth)
    uint64_t slice = nbytes / 16;
    if (slice < (8ull << 20)) slice = 8ull << 20;
    if (slice > (32ull << 20)) slice = 32ull << 20;
    slice = slice / fg_frame_slice_align() * fg_frame_slice_align();  // (whole 64 KiB tiles of the one-pass scan)
    const uint32_t slices = (uint32_t)((nbytes + slice - 1) / slice);
    TableShares shares(ctx, slices);
    const uint64_t nblk_total = nbytes / blk + 1;
    if ((rc = ensure_pipeline(ctx, slices)) != FG_OK) return rc;
    const hipStream_t s_up = ctx->s_up, s_run = ctx->s_run, s_down = ctx->s_down;
    auto drain = [&]() {
        (void)hipStreamSynchronize(s_up);
        (void)hipStreamSynchronize(s_run);
        (void)hipStreamSynchronize(s_down);
    };
    // capacities from the ctx's experience: frames, rows, entries
    const uint64_t cap = (uint64_t)((double)nbytes * ctx->frames_per_byte * 1.25) + 4096;
    const uint64_t ent_cap0 = fmt == FG_RFC3164 ? 16 : fmt == FG_RFC5424 ? nbytes / 16 + 1024 : nbytes / 8 + 1024;
    const uint64_t ent_cap = ent_cap0 > 0xFFFFFFF0ull ? 0xFFFFFFF0ull : ent_cap0;
    if ((rc = grow_dev(ctx, (void**)&ctx->d_bytes, &ctx->d_bytes_cap, up(nbytes, 16) + 16)) != FG_OK) return rc;
    if ((rc = grow_dev(ctx, (void**)&ctx->d_offsets, &ctx->d_offsets_cap, (cap + 2) * 8)) != FG_OK) return rc;
    if ((rc = grow_dev(ctx, (void**)&ctx->d_bad, &ctx->d_bad_cap, cap + 1)) != FG_OK) return rc;
    if ((rc = grow_dev(ctx, (void**)&ctx->d_frame, &ctx->d_frame_cap, fg_frame_scratch_bytes(nbytes))) != FG_OK) return rc;
    if ((rc = grow_pinned(ctx, (void**)&ctx->h_off, &ctx->h_off_cap, (cap + 2) * 8)) != FG_OK) return rc;
    if (!ctx->h_cnt) FG_HIP(ctx, hipHostMalloc((void**)&ctx->h_cnt, 65536, hipHostMallocDefault));
    if (slices + 2 > 4096) return FG_ERR_UNSUPPORTED;
    uint64_t* const cnt_dv = (uint64_t*)const_cast<void*>(device_view_of_pinned(ctx->h_cnt));
    if (!cnt_dv) return FG_ERR_UNSUPPORTED;
    uint64_t tab_bytes = 0;
    carve(nullptr, cap, ent_cap, nullptr, &tab_bytes);
    if ((rc = grow_pinned(ctx, (void**)&ctx->h_tab, &ctx->h_tab_cap, tab_bytes)) != FG_OK) return rc;
    fg_tables ht, kt;
    carve(ctx->h_tab, cap, ent_cap, &ht, nullptr);
    if ((rc = pinned_tables_for_kernels(ctx, ht, &kt)) != FG_OK) return rc;
    FG_HIP(ctx, hipMemsetAsync(kt.ent_used, 0, 8, s_run));
    FG_HIP(ctx, hipMemsetAsync(ctx->d_bad, 0, cap + 1, s_run));
    FG_HIP(ctx, hipEventRecord(ctx->ev_ready, s_run));
    FG_HIP(ctx, hipStreamWaitEvent(s_up, ctx->ev_ready, 0));  // (the framing kernels below write d_bad on the upload stream)
    const uint32_t delim = framing == FG_FRAME_LINE ? 0x0Au : 0x00u;
    std::vector<hipEvent_t>& ev = ctx->ev_slice;
    // A raw chunk in PINNED memory is not uploaded at all: the framing scan reads it where it is, over the link, and stores it to
    // HBM on the way (k_frame_scan<COPY>) -- the copy engine stays out of it, and the kernels of slice k + 1 (reads over the link, on
    // the upload stream) run beside the decode kernels of slice k (writes over the link).  Pageable memory: hipMemcpy per slice.
    // MEASURED (profiles/r04l_*): slower than the copy engine's uploads on this platform -- cfg2 91 vs 184 M lines/s, GELF 73 vs 96,
    // LTSV 84 vs 122 -- the framing kernels of slice k + 1 do not run beside the persistent decode grid of slice k, so the two
    // directions of the link take turns again.  Off by default (FG_LO_FRAME_KERNEL_UPLOAD turns it on for further work).
    const uint8_t* src_dv = ((ctx->lo.flags & FG_LO_FRAME_KERNEL_UPLOAD) && ((uintptr_t)bytes & 15u) == 0 && device_view_of_pinned(bytes + nbytes - 1))
                                ? (const uint8_t*)device_view_of_pinned(bytes) : nullptr;
    // Copy-engine uploads: EVERY upload is queued now, back to back on the upload stream (they depend on nothing: the link never waits
    // for the host, and nothing else is ever queued between two of them -- framing kernels on the same stream held the next upload
    // up for their whole duration: GELF 96 -> 69 M lines/s, profiles/r04y_e2e_*); slice k is framed on the kernel stream once its
    // event fires.  Kernel upload: the scan IS the upload, on the upload stream, a few slices ahead of the decodes.
    if (!src_dv) {
        for (uint32_t k = 0; k < slices; ++k) {
            const uint64_t b0 = (uint64_t)k * slice, b1 = k + 1 == slices ? nbytes : b0 + slice;
            FG_HIP(ctx, hipMemcpyAsync(ctx->d_bytes + b0, bytes + b0, b1 - b0, hipMemcpyHostToDevice, s_up));
            if (k + 1 == slices) FG_HIP(ctx, hipMemsetAsync(ctx->d_bytes + nbytes, 0, up(nbytes, 16) + 16 - nbytes, s_up));
            FG_HIP(ctx, hipEventRecord(ev[2 * k], s_up));
        }
    }
    // framing of slice k; its cumulative frame count lands in a pinned word (all asynchronous)
    auto enqueue = [&](uint32_t k) -> int {
        const uint64_t b0 = (uint64_t)k * slice, b1 = k + 1 == slices ? nbytes : b0 + slice;
        const hipStream_t s_frame = src_dv ? s_up : s_run;
        if (!src_dv) FG_HIP(ctx, hipStreamWaitEvent(s_run, ev[2 * k], 0));
        else if (k + 1 == slices) FG_HIP(ctx, hipMemsetAsync(ctx->d_bytes + up(nbytes, 16), 0, 16, s_up));
        uint64_t* d_total = nullptr;
        const uint64_t blk0 = b0 / blk, blk1 = k + 1 == slices ? nblk_total : b1 / blk;
        int lrc = fg_launch_frame_slice(ctx->d_bytes, nbytes, delim, ctx->d_frame, ctx->d_offsets, ctx->d_bad, cap, blk0, blk1, &d_total, s_frame, src_dv,
                                        (ctx->lo.flags & FG_LO_FRAME_CLASSIC) ? 1 : (ctx->lo.flags & FG_LO_FRAME_SELFTEST_STALL) ? 2 : 0);
        if (lrc == 0) lrc = fg_launch_poke64(d_total, cnt_dv + k, s_frame);
        if (lrc != 0) {
            ctx->last_hip = lrc;
            return FG_ERR_HIP;
        }
        FG_HIP(ctx, hipEventRecord(ev[2 * k + 1], s_frame));
        if (src_dv) FG_HIP(ctx, hipStreamWaitEvent(s_run, ev[2 * k + 1], 0));
        return FG_OK;
    };
    uint64_t done = 0;  // frames decoded so far
    auto decode_rows = [&](uint64_t f0, uint64_t f1, uint64_t span_bytes) -> int {  // frames [f0, f1): decode into the pinned tables
        if (f1 == f0) return FG_OK;
        const uint64_t rows = f1 - f0;
        fg_tables sl = kt;
        sl.n = rows;
        sl.meta += f0; sl.ts += f0; sl.hostname += f0; sl.appname += f0; sl.procid += f0; sl.msgid += f0; sl.msg += f0; sl.full_msg += f0;
        sl.ent_first += f0; sl.ent_count += f0;
        int r = fg_decode_frames_impl(ctx, fmt, framing, ctx->d_bytes, nbytes, ctx->d_offsets + f0, rows, ctx->d_bad + f0, &sl, (void*)s_run, false, span_bytes);
        if (r != FG_OK) return r;
        // the frames' offsets (8 bytes per frame) are the one thing still copied: behind the uploads on the copy engine, i.e. at the end
        FG_HIP(ctx, hipEventRecord(ctx->ev_ready, s_run));
        FG_HIP(ctx, hipStreamWaitEvent(s_down, ctx->ev_ready, 0));
        FG_HIP(ctx, hipMemcpyAsync(ctx->h_off + f0 + (f0 ? 1 : 0), ctx->d_offsets + f0 + (f0 ? 1 : 0), (rows + (f0 ? 0 : 1)) * 8, hipMemcpyDeviceToHost, s_down));
        return FG_OK;
    };
    uint32_t queued = 0;
    for (uint32_t k = 0; k < slices; ++k) {
        while (queued < slices && queued < k + (src_dv ? 4u : 2u)) {  // framing queued ahead of the decode being issued
            if ((rc = enqueue(queued)) != FG_OK) {
                drain();
                return rc;
            }
            ++queued;
        }
        if (hipEventSynchronize(ev[2 * k + 1]) != hipSuccess) {
            drain();
            return FG_ERR_HIP;
        }
        const uint64_t total = ctx->h_cnt[k];  // delimiters up to the end of slice k = frames that are complete
        if (total == FG_FRAME_ABORTED) {  // the one-pass scan gave up waiting on a tile (never seen): the one-piece path frames classically
            drain();
            return FG_ERR_UNSUPPORTED;
        }
        if (total + 1 > cap) {
            drain();
            ctx->frames_per_byte = (double)(total + 1) / (double)(((uint64_t)k + 1) * slice);
            return FG_ERR_UNSUPPORTED;
        }
        const uint64_t b1 = k + 1 == slices ? nbytes : ((uint64_t)k + 1) * slice;
        if (k + 1 < slices) {
            if ((rc = decode_rows(done, total, b1 - (uint64_t)k * slice)) != FG_OK) {
                drain();
                return rc;
            }
            done = total;
            continue;
        }
        // the last slice: an unterminated tail is one more frame when the stream ends here, else it stays with the caller
        uint64_t last_end = 0;
        FG_HIP(ctx, hipMemcpyAsync(&last_end, ctx->d_offsets + total, 8, hipMemcpyDeviceToHost, s_run));
        FG_HIP(ctx, hipStreamSynchronize(s_run));
        const bool tail = last_end != nbytes;
        uint64_t n = total;
        if (tail && final) {
            ctx->h_cnt[slices] = nbytes;
            FG_HIP(ctx, hipMemcpyAsync(ctx->d_offsets + total + 1, ctx->h_cnt + slices, 8, hipMemcpyHostToDevice, s_run));
            n = total + 1;
        }
        *consumed = (tail && !final) ? last_end : nbytes;
        if ((rc = decode_rows(done, n, *consumed > (uint64_t)k * slice ? *consumed - (uint64_t)k * slice : 1)) != FG_OK) {
            drain();
            return rc;
        }
        done = n;
    }
    {
        fg_tables all = kt;
        all.n = done;
        if ((rc = fg_finish_deferred_general(ctx, framing, ctx->d_bytes, ctx->d_offsets, done, ctx->d_bad, &all, (void*)s_run)) != FG_OK) {
            drain();
            return rc;
        }
    }
    uint64_t used = 0;
    FG_HIP(ctx, hipMemcpyAsync(&used, kt.ent_used, 8, hipMemcpyDeviceToHost, s_run));
    FG_HIP(ctx, hipStreamSynchronize(s_run));
    if (done) ctx->frames_per_byte = (double)done / (double)nbytes;
    drain();
    if (used > ent_cap) return FG_ERR_UNSUPPORTED;
    *ht.ent_used = used;
    ht.n = done;
    *out = ht;
    *out_offsets = ctx->h_off;
    *n_frames = done;
    if (done == 0) {
        fg_tables empty{};
        *out = empty;
    }
    return FG_OK;
}

// fg_frame_decode_batch as ONE LAUNCH (round 6): a raw chunk in PINNED memory is neither uploaded nor framed by a pass of its own -- the
// decode kernel reads it in place over the link, frames its tiles itself (fg_fused.hpp) and writes rows, entries AND the frame offsets
// straight into the ctx's pinned tables: what fg_decode_batch's zero-copy form does for framed lines, now for the raw stream -- the path a
// flowgger input really has (LineSplitter::run over a socket's bytes).  Until round 5 this path was an upload + a framing scan + a capped
// decode grid per slice and ran at 0.55-0.87 of the link where the zero-copy form did 0.80-0.97.
// FG_ERR_UNSUPPORTED: the chunk does not qualify (pageable memory, long lines, a table estimate that turned out too small, the
// look-back's bounded wait ran out): the caller takes the paths below.
static int frame_decode_fused(fg_ctx* ctx, fg_format fmt, fg_framing framing, const uint8_t* bytes, uint64_t nbytes, int final, fg_tables* out,
                              const uint64_t** out_offsets, uint64_t* n_frames, uint64_t* consumed) {
    if (nbytes == 0 || (ctx->lo.flags & (FG_LO_NO_FUSED_FRAMING | FG_LO_NO_ZERO_COPY | FG_LO_FRAME_CLASSIC | FG_LO_FRAME_SELFTEST_STALL | FG_LO_FRAME_KERNEL_UPLOAD))) return FG_ERR_UNSUPPORTED;
    if (fmt != FG_RFC5424 && fmt != FG_LTSV && fmt != FG_GELF) return FG_ERR_UNSUPPORTED;
    const uint8_t* d_bytes = (const uint8_t*)device_view_of_pinned(bytes);
    if (!d_bytes || ((uintptr_t)d_bytes & 15u) != 0) return FG_ERR_UNSUPPORTED;
    {
        const uint8_t* last_b = bytes + ((nbytes + 15u) & ~15ull) - 1u;
        const uint8_t* dv_last = (const uint8_t*)device_view_of_pinned(last_b);
        if (!dv_last || dv_last - d_bytes != last_b - bytes) return FG_ERR_UNSUPPORTED;  // (ONE mapping up to the last byte the kernels load)
    }
    LinkBoundGrid grid(ctx);
    int rc;
    // tables from the ctx's experience (frames per byte of its last chunk) and one entry per 16 / 8 input bytes; a chunk that holds more
    // of either says how many, and the launch is repeated ONCE PER TABLE with what it needs (another pass over the link -- the price
    // of the older forms' count-first as well -- and the next chunk is sized from this one)
    uint64_t cap = (uint64_t)((double)nbytes * ctx->frames_per_byte * 1.25) + 4096;
    uint64_t ent_cap = fmt == FG_RFC5424 ? nbytes / 16 + 1024 : nbytes / 8 + 1024;
    if (!ctx->h_cnt) FG_HIP(ctx, hipHostMalloc((void**)&ctx->h_cnt, 65536, hipHostMallocDefault));
    const hipStream_t s = ctx->stream;
    fg_tables ht;
    uint64_t total = 0, used = 0;
    for (int attempt = 0;; ++attempt) {
        if (ent_cap > 0xFFFFFFF0ull) ent_cap = 0xFFFFFFF0ull;
        uint64_t tab_bytes = 0;
        carve(nullptr, cap, ent_cap, nullptr, &tab_bytes);
        if ((rc = grow_pinned(ctx, (void**)&ctx->h_tab, &ctx->h_tab_cap, tab_bytes)) != FG_OK) return rc;
        if ((rc = grow_pinned(ctx, (void**)&ctx->h_off, &ctx->h_off_cap, (cap + 2) * 8)) != FG_OK) return rc;
        fg_tables kt;
        carve(ctx->h_tab, cap, ent_cap, &ht, nullptr);
        if ((rc = pinned_tables_for_kernels(ctx, ht, &kt)) != FG_OK) return rc;
        uint64_t* const off_dv = (uint64_t*)const_cast<void*>(device_view_of_pinned(ctx->h_off));
        if (!off_dv) return FG_ERR_UNSUPPORTED;
        unsigned long long* d_total = nullptr;
        if ((rc = fg_frame_decode_impl(ctx, fmt, framing, d_bytes, nbytes, final, off_dv, cap, &kt, 0, (void*)s, &d_total)) != FG_OK) return rc;
        uint64_t* const res = ctx->h_cnt + 2048;  // [0] frames [1] abort [2] entries
        FG_HIP(ctx, hipMemcpyAsync(res, d_total, 16, hipMemcpyDeviceToHost, s));
        FG_HIP(ctx, hipMemcpyAsync(res + 2, kt.ent_used, 8, hipMemcpyDeviceToHost, s));
        FG_HIP(ctx, hipStreamSynchronize(s));
        total = res[0], used = res[2];
        if (res[1] != 0) return FG_ERR_UNSUPPORTED;  // (the look-back's bounded wait ran out: never seen)
        const bool rows_short = total > cap, ents_short = used > ent_cap;
        if (!rows_short && !ents_short) break;
        if (rows_short) ctx->frames_per_byte = (double)(total + 1) / (double)nbytes;
        if (attempt >= 2 || (ents_short && ent_cap >= 0xFFFFFFF0ull)) return FG_ERR_UNSUPPORTED;
        if (rows_short) cap = total + total / 16 + 1024;
        if (ents_short) ent_cap = used + used / 8 + 1024;  // (rows beyond the old capacity had no entries yet: a third launch may follow)
    }
    if (total && nbytes >= (1u << 20)) ctx->frames_per_byte = (double)total / (double)nbytes;
    *n_frames = total;
    ctx->last_host_path = FG_PATH_FRAME_FUSED;
    if (total == 0) {
        fg_tables empty{};
        *out = empty;
        ctx->h_off[0] = 0;
        *out_offsets = ctx->h_off;
        *consumed = 0;
        return FG_OK;
    }
    *ht.ent_used = used;
    ht.n = total;
    *out = ht;
    *out_offsets = ctx->h_off;
    *consumed = ctx->h_off[total];
    return FG_OK;
}

int fg_frame_decode_batch(fg_ctx* ctx, fg_format fmt, fg_framing framing, const uint8_t* bytes, uint64_t nbytes, int final,
                          fg_tables* out, const uint64_t** out_offsets, uint64_t* n_frames, uint64_t* consumed) {
    if (!ctx || !out || !out_offsets || !n_frames || !consumed || (nbytes && !bytes)) return FG_ERR_ARG;
    if (framing != FG_FRAME_LINE && framing != FG_FRAME_NUL) return FG_ERR_UNSUPPORTED;
    {
        DeviceGuard g(ctx->device);
        *n_frames = 0;
        *consumed = 0;
        *out_offsets = nullptr;
        const int rc = frame_decode_fused(ctx, fmt, framing, bytes, nbytes, final, out, out_offsets, n_frames, consumed);
        if (rc != FG_ERR_UNSUPPORTED) return rc;
    }
    if (nbytes >= (48ull << 20) && !(ctx->lo.flags & FG_LO_TRANSCODE_ONE_PIECE)) {
        DeviceGuard g(ctx->device);
        *n_frames = 0;
        *consumed = 0;
        *out_offsets = nullptr;
        const int rc = frame_decode_sliced(ctx, fmt, framing, bytes, nbytes, final, out, out_offsets, n_frames, consumed);
        if (rc == FG_OK) ctx->last_host_path = FG_PATH_FRAME_SLICED;
        if (rc != FG_ERR_UNSUPPORTED) return rc;
    }
    const int rc1 = frame_decode_one_piece(ctx, fmt, framing, bytes, nbytes, final, out, out_offsets, n_frames, consumed);
    if (rc1 == FG_OK) ctx->last_host_path = FG_PATH_FRAME_ONE_PIECE;
    return rc1;
}

// Framing stage shared by fg_frame_decode_batch and fg_transcode_batch: the raw chunk is already in ctx->d_bytes
// (zero padded); fills ctx->d_offsets / ctx->d_bad and says how many frames the chunk holds and how many of its bytes
// they cover (an unterminated tail is a frame only when `final`).
static int frame_stage(fg_ctx* ctx, fg_framing framing, uint64_t nbytes, int final, uint64_t* n_frames, uint64_t* consumed) {
    hipStream_t s = ctx->stream;
    int rc;
    uint64_t cap = nbytes / 32 + 1024, total = 0, last_end = 0;
    for (;;) {  // capacity: one frame per 32 bytes to start with, exact on retry
        if ((rc = grow_dev(ctx, (void**)&ctx->d_offsets, &ctx->d_offsets_cap, (cap + 2) * 8)) != FG_OK) return rc;
        if ((rc = grow_dev(ctx, (void**)&ctx->d_bad, &ctx->d_bad_cap, cap + 1)) != FG_OK) return rc;
        if ((rc = grow_dev(ctx, (void**)&ctx->d_frame, &ctx->d_frame_cap, fg_frame_scratch_bytes(nbytes))) != FG_OK) return rc;
        for (int classic = (ctx->lo.flags & FG_LO_FRAME_CLASSIC) ? 1 : (ctx->lo.flags & FG_LO_FRAME_SELFTEST_STALL) ? 2 : 0;; classic = 1) {
            uint64_t* d_total = nullptr;
            int lrc = fg_launch_frame(ctx->d_bytes, nbytes, framing == FG_FRAME_LINE ? 0x0Au : 0x00u, ctx->d_frame, ctx->d_offsets,
                                      ctx->d_bad, cap, &d_total, s, classic);
            if (lrc != 0) {
                ctx->last_hip = lrc;
                return FG_ERR_HIP;
            }
            FG_HIP(ctx, hipMemcpyAsync(&total, d_total, 8, hipMemcpyDeviceToHost, s));
            FG_HIP(ctx, hipStreamSynchronize(s));
            if (total != FG_FRAME_ABORTED || classic == 1) break;  // (the one-pass scan gave up waiting on a tile: the three-kernel form)
        }
        if (total + 1 <= cap) break;
        cap = total + 16;
    }
    FG_HIP(ctx, hipMemcpyAsync(&last_end, ctx->d_offsets + total, 8, hipMemcpyDeviceToHost, s));
    FG_HIP(ctx, hipStreamSynchronize(s));
    const bool tail = last_end != nbytes;  // terminated frames = total
    *n_frames = total + ((tail && final) ? 1 : 0);
    *consumed = (tail && !final) ? last_end : nbytes;
    return FG_OK;
}

// Decode ctx->d_bytes / ctx->d_offsets into tables carved from ctx->d_tab, growing the entry table until it fits.
static int decode_stage(fg_ctx* ctx, fg_format fmt, fg_framing framing, uint64_t nbytes, uint64_t n, const uint8_t* d_bad,
                        fg_tables* dt, uint64_t* ent_used) {
    hipStream_t s = ctx->stream;
    int rc;
    uint64_t ent_cap = fmt == FG_RFC5424 ? nbytes / 16 + 1024 : nbytes / 8 + 1024;
    if (fmt == FG_RFC3164) ent_cap = 16;  // RFC3164 produces no entries
    for (;;) {
        if (ent_cap > 0xFFFFFFF0ull) ent_cap = 0xFFFFFFF0ull;
        uint64_t bytes_total = 0;
        carve(nullptr, n, ent_cap, nullptr, &bytes_total);
        if ((rc = grow_dev(ctx, (void**)&ctx->d_tab, &ctx->d_tab_cap, bytes_total)) != FG_OK) return rc;
        carve(ctx->d_tab, n, ent_cap, dt, nullptr);
        rc = fg_decode_frames_device(ctx, fmt, framing, ctx->d_bytes, nbytes, ctx->d_offsets, n, d_bad, dt, FG_STREAM_OWN);
        if (rc != FG_OK) return rc;
        uint64_t used = 0;
        FG_HIP(ctx, hipMemcpyAsync(&used, dt->ent_used, 8, hipMemcpyDeviceToHost, s));
        FG_HIP(ctx, hipStreamSynchronize(s));
        if (used > ent_cap) {
            if (ent_cap >= 0xFFFFFFF0ull) return FG_ERR_ENT_OVERFLOW;
            ent_cap = used + used / 8 + 1024;
            continue;
        }
        *ent_used = used;
        return FG_OK;
    }
}

// fg_transcode_batch for a LARGE batch of framed lines: the batch goes through H2D -> decode -> encode -> D2H as slices of ~32 MiB
// on the ctx's two streams, so that the upload of slice k+1 and the (2-3x larger) download of slice k's messages share the
// full-duplex link, and the kernels hide behind both.  The host only waits for one small number per slice (its encoded size:
// the next slice's messages start there).  Returns FG_ERR_UNSUPPORTED when the batch must take the one-piece path (an output
// estimate that turned out too small): nothing has been returned to the caller by then.
static int transcode_sliced(fg_ctx* ctx, fg_format fmt, const fg_encode_cfg* ecfg, const uint8_t* bytes, uint64_t nbytes,
                            const uint64_t* offsets, uint64_t n, fg_transcoded* out) {
    int rc;
    if (n < 64) return FG_ERR_UNSUPPORTED;
    // Slices that GROW: 4, 8, 16, 32 MiB, then 64 MiB each (a sixtieth of the batch beyond 4 GiB).  The download direction is the bound (the
    // GELF text is 2.4x the input) and it cannot start before the first slice is uploaded, decoded, counted, scanned and written:
    // 1.7 ms of a 13 ms call with a 32 MiB first slice, a quarter of a millisecond with 4 MiB.
    std::vector<uint64_t> cut{0};
    {
        const uint64_t steady = nbytes / 60u > (64ull << 20) ? nbytes / 60u : (64ull << 20);
        uint64_t piece = 4ull << 20, at = offsets[0];
        while (cut.back() < n) {
            at += piece;
            uint64_t line = (uint64_t)(std::upper_bound(offsets, offsets + n + 1, at) - offsets);
            line = line ? line - 1 : 0;  // the last line that starts at or before `at`
            if (line <= cut.back()) line = cut.back() + 1;
            if (line >= n || offsets[n] - offsets[line] < piece / 2u) line = n;
            cut.push_back(line);
            at = offsets[line];
            if (piece < steady) piece = piece * 2u < steady ? piece * 2u : steady;
        }
    }
    const uint32_t slices = (uint32_t)(cut.size() - 1);
    // One stream per direction of the link (round 6; rounds 3-5 ran two lanes, each with its own upload, kernels and download).  The
    // timeline of one call (profiles/r06ae_transcode_timeline.json) showed what two lanes cost: whenever both lanes had a copy of the
    // SAME direction in flight the runtime moved the second with a blit KERNEL, copies and kernels of a lane waited for each other, and
    // the two directions were never busy together.  Now:
    //   s_up    upload k, decode k, count k, upload k + 1, ... -- all queued at once, nothing on it waits for the host; the upload
    //           direction has time to spare (the text is 2.4x the input), so the kernels may sit between two uploads
    //   s_run   what hangs on the host's one number per slice: scan k -> the slice's size -> write k.  A write never queues behind the
    //           decode of a later slice: kernels crawl while a download saturates the link's write queue (k_rfc5424 0.03 -> 0.6-1.0 ms
    //           per slice), and only the write is on the download's critical path
    //   s_down  every download, in slice order
    // (three streams, not four: with a fourth the runtime put it on the hardware queue of s_down and every download waited for the
    //  last upload -- profiles/r06ai_transcode_timeline.json)
    if ((rc = ensure_pipeline(ctx, slices * 2u)) != FG_OK) return rc;  // (four events per slice)
    const hipStream_t s_up = ctx->s_up, s_run = ctx->s_run, s_down = ctx->s_down, s_ahead = ctx->s_up;
    hipStream_t lanes[1] = {s_run};
    if (!ctx->h_cnt) FG_HIP(ctx, hipHostMalloc((void**)&ctx->h_cnt, 65536, hipHostMallocDefault));
    std::vector<hipEvent_t>& ev = ctx->ev_slice;  // [4k + 1] slice k is decoded and counted, [4k + 2] its messages are written
    TableShares shares(ctx, slices);
    // ---- device buffers: input, tables (whole batch), out_offsets / enc_status, encoder scratch ----
    if ((rc = grow_dev(ctx, (void**)&ctx->d_bytes, &ctx->d_bytes_cap, up(nbytes, 16) + 16)) != FG_OK) return rc;
    if ((rc = grow_dev(ctx, (void**)&ctx->d_offsets, &ctx->d_offsets_cap, (n + 1) * 8)) != FG_OK) return rc;
    uint64_t ent_cap = fmt == FG_RFC5424 ? nbytes / 16 + 1024 : nbytes / 8 + 1024;
    if (fmt == FG_RFC3164) ent_cap = 16;
    if (ent_cap > 0xFFFFFFF0ull) ent_cap = 0xFFFFFFF0ull;
    uint64_t tab_bytes = 0;
    carve(nullptr, n, ent_cap, nullptr, &tab_bytes);
    if ((rc = grow_dev(ctx, (void**)&ctx->d_tab, &ctx->d_tab_cap, tab_bytes)) != FG_OK) return rc;
    fg_tables dt{};
    carve(ctx->d_tab, n, ent_cap, &dt, nullptr);
    const uint64_t offs_bytes = up((n + 1) * 8, 256);
    if ((rc = grow_dev(ctx, (void**)&ctx->d_tmeta, &ctx->d_tmeta_cap, offs_bytes + up(n, 256))) != FG_OK) return rc;
    uint64_t* d_out_offsets = reinterpret_cast<uint64_t*>(ctx->d_tmeta);
    uint8_t* d_enc_status = ctx->d_tmeta + offs_bytes;
    fg::EncCfgHost h;
    if (!fg::build_enc_cfg(fmt, ecfg, ctx->suffix, ctx->has_suffix, &h)) return FG_ERR_ARG;
    const uint64_t keys_bytes = up(h.keys.size() * sizeof(fg::StaticKey), 16), blob_bytes = up(h.blob.size() + 16, 256);
    const uint64_t cfg_bytes = up(keys_bytes + blob_bytes, 256);
    const uint64_t sizes_bytes = up(n * 4 + 4, 256), sums_bytes = up((n / 64 + slices + 2) * 8, 256);
    if ((rc = grow_dev(ctx, (void**)&ctx->d_enc, &ctx->d_enc_cap, cfg_bytes + sizes_bytes + sums_bytes)) != FG_OK) return rc;
    std::vector<uint8_t> host(cfg_bytes, 0);
    if (!h.keys.empty()) memcpy(host.data(), h.keys.data(), h.keys.size() * sizeof(fg::StaticKey));
    if (!h.blob.empty()) memcpy(host.data() + keys_bytes, h.blob.data(), h.blob.size());
    FG_HIP(ctx, hipMemcpyAsync(ctx->d_enc, host.data(), host.size(), hipMemcpyHostToDevice, lanes[0]));
    FG_HIP(ctx, hipMemcpyAsync(ctx->d_offsets, offsets, (n + 1) * 8, hipMemcpyHostToDevice, lanes[0]));
    FG_HIP(ctx, hipMemsetAsync(dt.ent_used, 0, 8, lanes[0]));
    FG_HIP(ctx, hipStreamSynchronize(lanes[0]));  // (`host` is a local; lane 1 may start)
    fg::EncCfg cfg = h.cfg;
    cfg.keys = reinterpret_cast<const fg::StaticKey*>(ctx->d_enc);
    cfg.blob = ctx->d_enc + keys_bytes;
    const uint32_t cfg_lds = keys_bytes + h.blob.size() <= 4096 ? (uint32_t)up(keys_bytes + h.blob.size(), 16) : 0u;
    uint32_t* d_sizes = reinterpret_cast<uint32_t*>(ctx->d_enc + cfg_bytes);
    uint64_t* d_block_sums = reinterpret_cast<uint64_t*>(ctx->d_enc + cfg_bytes + sizes_bytes);
    const uint32_t tile_cap = pick_tile_cap(ctx, nbytes, n, 40 * 1024, 1);
    fg::DevTables ddt = to_dev(dt);
    // ---- host buffer: fixed-size arrays first, the messages behind them (they grow slice by slice) ----
    const uint64_t o_offs = 0, o_meta = o_offs + offs_bytes, o_st = o_meta + up(n * 4, 256), o_msgs = o_st + up(n, 256);
    uint64_t base = 0;  // encoded bytes of the slices finished so far
    auto sl_tables = [&](uint64_t l0, uint64_t l1) {
        fg_tables sl = dt;
        sl.n = l1 - l0;
        sl.meta += l0; sl.ts += l0; sl.hostname += l0; sl.appname += l0; sl.procid += l0; sl.msgid += l0; sl.msg += l0; sl.full_msg += l0;
        sl.ent_first += l0; sl.ent_count += l0;
        return sl;
    };
    auto blocks_before = [&](uint32_t k) { return cut[k] / 64 + k; };  // first scratch sum of slice k (disjoint per slice)
    auto drain = [&]() {
        (void)hipStreamSynchronize(s_up);
        (void)hipStreamSynchronize(s_run);
        (void)hipStreamSynchronize(s_down);
    };
    // queue a slice's upload, decode and count kernel (sixteen-byte aligned cuts: a slice starts with the sixteen bytes its first line begins in)
    auto enqueue = [&](uint32_t k) -> int {
        hipStream_t s = s_ahead;
        const uint64_t l0 = cut[k], l1 = cut[k + 1];
        if (l1 == l0) return FG_OK;
        const uint64_t b0 = offsets[l0] & ~15ull, b1 = offsets[l1];
        if (b1 > b0) FG_HIP(ctx, hipMemcpyAsync(ctx->d_bytes + b0, bytes + b0, b1 - b0, hipMemcpyHostToDevice, s));
        const fg_tables sl = sl_tables(l0, l1);
        // (the two lanes' kernels may be in flight at the same time: each lane parks its entries in its own stash)
        int r = fg_decode_frames_impl(ctx, fmt, FG_FRAME_NONE, ctx->d_bytes, nbytes, ctx->d_offsets + l0, l1 - l0, nullptr, &sl, (void*)s, false,
                                   offsets[l1] - offsets[l0], k & 1u);
        if (r != FG_OK) return r;
        if (k == 0 && ecfg->encoder == FG_ENC_GELF) {  // the GELF ranking scratch is sized by the pairs per line: look at the first slice
            uint64_t used = ~0ull;
            FG_HIP(ctx, hipMemcpyAsync(&used, dt.ent_used, 8, hipMemcpyDeviceToHost, s));
            FG_HIP(ctx, hipStreamSynchronize(s));
            if (used == 0) cfg.sort_slots = 1;
            else if (used <= 2 * (l1 - l0)) cfg.sort_slots = 8;
        }
        fg::DevTables sdt = to_dev(sl);
        sdt.ent_cap = ddt.ent_cap;
        if (fg_launch_encode_count(ctx->d_bytes, ctx->d_offsets + l0, l1 - l0, &sdt, &cfg, tile_cap, cfg_lds, d_sizes + l0,
                                   d_block_sums + blocks_before(k), d_enc_status + l0, s) != 0)
            return FG_ERR_HIP;
        FG_HIP(ctx, hipEventRecord(ev[4 * k + 1], s));
        return FG_OK;
    };
    // (a few slices ahead of the one being finished, not all at once: the forty API calls of a slice take the host a tenth of a
    //  millisecond, and thirty slices queued before the first scan kept the first download waiting for four)
    uint32_t queued = 0;
    for (uint32_t k = 0; k < slices; ++k) {
        while (queued < slices && queued < k + 3u) {
            if ((rc = enqueue(queued)) != FG_OK) {
                drain();
                return rc;
            }
            ++queued;
        }

        // ---- finish slice k: offsets from `base`, its size (the host waits for that one number), the write kernel, the download ----
        hipStream_t s = s_run;
        const uint64_t l0 = cut[k], l1 = cut[k + 1], rows = l1 - l0;
        if (rows == 0) continue;
        if (hipStreamWaitEvent(s, ev[4 * k + 1], 0) != hipSuccess) {
            drain();
            return FG_ERR_HIP;
        }
        if (fg_launch_encode_scan(d_sizes + l0, d_block_sums + blocks_before(k), rows, d_out_offsets + l0, base, s) != 0) {
            drain();
            return FG_ERR_HIP;
        }
        uint64_t* const h_end = ctx->h_cnt + 4096;
        if (hipMemcpyAsync(h_end, d_out_offsets + l1, 8, hipMemcpyDeviceToHost, s) != hipSuccess || hipEventRecord(ctx->ev_ready, s) != hipSuccess) {
            drain();
            return FG_ERR_HIP;
        }
        if (hipEventSynchronize(ctx->ev_ready) != hipSuccess) {
            drain();
            return FG_ERR_HIP;
        }
        const uint64_t end = *h_end;
        if (k == 0) {
            // size the output buffers from the first slice (+ 12 %); a batch that outgrows the estimate takes the one-piece path
            const uint64_t in0 = offsets[l1] - offsets[l0];
            const uint64_t est = (uint64_t)((double)end * ((double)nbytes / (double)(in0 ? in0 : 1)) * 1.12) + (4ull << 20);
            if ((rc = grow_dev(ctx, (void**)&ctx->d_tout, &ctx->d_tout_cap, est)) != FG_OK || (rc = grow_pinned(ctx, (void**)&ctx->h_tout, &ctx->h_tout_cap, o_msgs + est)) != FG_OK) {
                drain();
                return rc;
            }
        }
        if (end > ctx->d_tout_cap || o_msgs + end > ctx->h_tout_cap) {
            drain();
            return FG_ERR_UNSUPPORTED;
        }
        const fg_tables sl = sl_tables(l0, l1);
        fg::DevTables sdt = to_dev(sl);
        sdt.ent_cap = ddt.ent_cap;
        if (fg_launch_encode_write(ctx->d_bytes, ctx->d_offsets + l0, rows, &sdt, &cfg, tile_cap, cfg_lds, d_out_offsets + l0, ctx->d_tout, d_sizes + l0, s) != 0) {
            drain();
            return FG_ERR_HIP;
        }
        uint8_t* hh = ctx->h_tout;
        // (the downloads stay with hipMemcpyAsync: a copy kernel of the library's own through the buffer's device view -- one wave per
        //  CU, eight 16-byte accesses in flight per lane -- moved a slice no faster than the runtime's blit kernel, 50 vs 52 GB/s, and
        //  slowed the kernels beside it just the same: it is the link's write queue that holds them up, not the CUs the copy takes)
        // at most three downloads queued: a stream's packets share a hardware queue with another stream's, and two dozen downloads
        // queued ahead held the uploads behind them up for 45 ms (4 M lines, profiles/r06al_transcode_host_trace.log)
        bool ok = k < 3u || hipEventSynchronize(ev[4 * (k - 3u) + 3]) == hipSuccess;
        ok = ok && hipEventRecord(ev[4 * k + 2], s) == hipSuccess && hipStreamWaitEvent(s_down, ev[4 * k + 2], 0) == hipSuccess;
        if (ok && end > base) ok = hipMemcpyAsync(hh + o_msgs + base, ctx->d_tout + base, end - base, hipMemcpyDeviceToHost, s_down) == hipSuccess;
        ok = ok && hipEventRecord(ev[4 * k + 3], s_down) == hipSuccess;
        if (!ok) {
            drain();
            return FG_ERR_HIP;
        }
        base = end;
    }
    // the fixed-size arrays -- offsets, meta, status of every line -- in three copies behind the last slice's messages (per slice they
    // were three more launches between two downloads: a tenth of a millisecond of an idle link per slice)
    {
        uint8_t* hh = ctx->h_tout;
        bool ok = hipStreamWaitEvent(s_down, ev[4 * (slices - 1u) + 2], 0) == hipSuccess;
        ok = ok && hipMemcpyAsync(hh + o_offs, d_out_offsets, (n + 1) * 8, hipMemcpyDeviceToHost, s_down) == hipSuccess;
        ok = ok && hipMemcpyAsync(hh + o_meta, dt.meta, n * 4, hipMemcpyDeviceToHost, s_down) == hipSuccess;
        ok = ok && hipMemcpyAsync(hh + o_st, d_enc_status, n, hipMemcpyDeviceToHost, s_down) == hipSuccess;
        if (!ok) {
            drain();
            return FG_ERR_HIP;
        }
    }
    drain();
    // an entry table that was too small shows up as FG_ST_OVERFLOW rows: the one-piece path sizes it exactly
    uint64_t used = 0;
    FG_HIP(ctx, hipMemcpy(&used, dt.ent_used, 8, hipMemcpyDeviceToHost));
    if (used > ent_cap) return FG_ERR_UNSUPPORTED;
    uint8_t* hh = ctx->h_tout;
    out->n = n;
    out->consumed = nbytes;
    out->out = hh + o_msgs;
    out->out_bytes = base;
    out->out_offsets = reinterpret_cast<const uint64_t*>(hh + o_offs);
    out->meta = reinterpret_cast<const uint32_t*>(hh + o_meta);
    out->enc_status = hh + o_st;
    out->frame_offsets = nullptr;
    return FG_OK;
}

int fg_transcode_batch(fg_ctx* ctx, fg_format fmt, fg_framing framing, const fg_encode_cfg* ecfg, const uint8_t* bytes,
                       uint64_t nbytes, const uint64_t* offsets, uint64_t n, int final, fg_transcoded* out) {
    if (!ctx || !ecfg || !out || (nbytes && !bytes)) return FG_ERR_ARG;
    if ((int)framing < 0 || (int)framing > 2) return FG_ERR_ARG;
    if (framing == FG_FRAME_NONE) {
        if (n && !offsets) return FG_ERR_ARG;
        if (n && (offsets[n] > nbytes || offsets[0] > offsets[n])) return FG_ERR_ARG;
    } else if (offsets) {
        return FG_ERR_ARG;  // a raw stream chunk is framed here; it does not come with offsets
    }
    *out = fg_transcoded{};
    if (framing != FG_FRAME_NONE && nbytes == 0) return FG_OK;
    DeviceGuard g(ctx->device);
    hipStream_t s = ctx->stream;
    int rc;
    // (from 16 MiB: the slices start at 4 MiB, so a 25 MB batch already has its first download in flight while the rest uploads --
    //  100 000 lines 1.65 vs 1.81 ms in one piece, 250 000 lines 3.35 vs 4.2; at 13 MB the two forms are level)
    if (framing == FG_FRAME_NONE && nbytes >= (16ull << 20) && n >= 4096 && !(ctx->lo.flags & FG_LO_TRANSCODE_ONE_PIECE)) {
        rc = transcode_sliced(ctx, fmt, ecfg, bytes, nbytes, offsets, n, out);
        if (rc != FG_ERR_UNSUPPORTED) return rc;
        *out = fg_transcoded{};
    }
    // 1. the chunk (and, for framed input, its offsets) to HBM
    if ((rc = grow_dev(ctx, (void**)&ctx->d_bytes, &ctx->d_bytes_cap, up(nbytes, 16) + 16)) != FG_OK) return rc;
    if (nbytes) FG_HIP(ctx, hipMemcpyAsync(ctx->d_bytes, bytes, nbytes, hipMemcpyHostToDevice, s));
    FG_HIP(ctx, hipMemsetAsync(ctx->d_bytes + nbytes, 0, up(nbytes, 16) + 16 - nbytes, s));
    uint64_t consumed = nbytes;
    const uint8_t* d_bad = nullptr;
    if (framing == FG_FRAME_NONE) {
        if ((rc = grow_dev(ctx, (void**)&ctx->d_offsets, &ctx->d_offsets_cap, (n + 1) * 8)) != FG_OK) return rc;
        if (n) FG_HIP(ctx, hipMemcpyAsync(ctx->d_offsets, offsets, (n + 1) * 8, hipMemcpyHostToDevice, s));
    } else {
        if ((rc = frame_stage(ctx, framing, nbytes, final, &n, &consumed)) != FG_OK) return rc;
        d_bad = ctx->d_bad;
    }
    out->n = n;
    out->consumed = consumed;
    if (n == 0) {
        FG_HIP(ctx, hipStreamSynchronize(s));
        return FG_OK;
    }
    // 2. decode (tables stay in HBM)
    fg_tables dt{};
    uint64_t ent_used = 0;
    if ((rc = decode_stage(ctx, fmt, framing, consumed, n, d_bad, &dt, &ent_used)) != FG_OK) return rc;
    // 3. encode + frame from the tables; the output buffer grows to the batch (steady state: one count + one write)
    const uint64_t offs_bytes = up((n + 1) * 8, 256);
    if ((rc = grow_dev(ctx, (void**)&ctx->d_tmeta, &ctx->d_tmeta_cap, offs_bytes + up(n, 256))) != FG_OK) return rc;
    uint64_t* d_out_offsets = reinterpret_cast<uint64_t*>(ctx->d_tmeta);
    uint8_t* d_enc_status = ctx->d_tmeta + offs_bytes;
    if (!ctx->d_tout && (rc = grow_dev(ctx, (void**)&ctx->d_tout, &ctx->d_tout_cap, consumed + consumed / 2 + 4096)) != FG_OK) return rc;
    uint64_t total = 0;
    rc = fg_encode_device(ctx, fmt, ecfg, ctx->d_bytes, consumed, ctx->d_offsets, n, &dt, ctx->d_tout, ctx->d_tout_cap, d_out_offsets,
                          d_enc_status, &total, FG_STREAM_OWN);
    if (rc == FG_ERR_ENT_OVERFLOW) {
        if ((rc = grow_dev(ctx, (void**)&ctx->d_tout, &ctx->d_tout_cap, total + 4096)) != FG_OK) return rc;
        rc = fg_encode_device(ctx, fmt, ecfg, ctx->d_bytes, consumed, ctx->d_offsets, n, &dt, ctx->d_tout, ctx->d_tout_cap, d_out_offsets,
                              d_enc_status, &total, FG_STREAM_OWN);
    }
    if (rc != FG_OK) return rc;
    // 4. only the encoded stream and the per-line verdicts cross PCIe back
    const uint64_t o_msgs = 0, o_offs = up(total, 256), o_meta = o_offs + offs_bytes, o_st = o_meta + up(n * 4, 256),
                   o_frames = o_st + up(n, 256), h_total = o_frames + (framing != FG_FRAME_NONE ? offs_bytes : 0);
    if ((rc = grow_pinned(ctx, (void**)&ctx->h_tout, &ctx->h_tout_cap, h_total)) != FG_OK) return rc;
    uint8_t* h = ctx->h_tout;
    if (total) FG_HIP(ctx, hipMemcpyAsync(h + o_msgs, ctx->d_tout, total, hipMemcpyDeviceToHost, s));
    FG_HIP(ctx, hipMemcpyAsync(h + o_offs, d_out_offsets, (n + 1) * 8, hipMemcpyDeviceToHost, s));
    FG_HIP(ctx, hipMemcpyAsync(h + o_meta, dt.meta, n * 4, hipMemcpyDeviceToHost, s));
    FG_HIP(ctx, hipMemcpyAsync(h + o_st, d_enc_status, n, hipMemcpyDeviceToHost, s));
    if (framing != FG_FRAME_NONE) FG_HIP(ctx, hipMemcpyAsync(h + o_frames, ctx->d_offsets, (n + 1) * 8, hipMemcpyDeviceToHost, s));
    FG_HIP(ctx, hipStreamSynchronize(s));
    out->out = h + o_msgs;
    out->out_bytes = total;
    out->out_offsets = reinterpret_cast<const uint64_t*>(h + o_offs);
    out->meta = reinterpret_cast<const uint32_t*>(h + o_meta);
    out->enc_status = h + o_st;
    out->frame_offsets = framing != FG_FRAME_NONE ? reinterpret_cast<const uint64_t*>(h + o_frames) : nullptr;
    return FG_OK;
}

}  // extern "C"
