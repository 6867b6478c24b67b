// fg_ctx.hpp -- private to the host side of the C ABI (fg_capi.cpp: contexts + the device entry points; fg_host_pipeline.cpp: the
// PCIe-inclusive host-buffer entry points): the ctx, the kernel launchers' prototypes and the small helpers both translation units use.
// Not installed, not part of the ABI (include/fg_hip.h is).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <string>
#include <vector>

#include "../../include/fg_hip.h"
#include "fg_device.hpp"
#include "fg_enc_cfg.hpp"
#include "fg_fused_plan.hpp"
#include <time.h>

#include "fg_rfc3164_parse.hpp"
#include "fg_tz_index.hpp"

namespace fg {
// device view of input.ltsv_schema / input.ltsv_suffixes (must match fg_ltsv.hip)
struct LtsvDevCfg {
    uint32_t n_schema;
    const uint8_t* blob;
    const uint32_t* name_off;
    const uint8_t* types;
    uint32_t suf_off[4];
    uint32_t suf_len[4];
    uint32_t has_suf[4];
};
}  // namespace fg

extern "C" int fg_launch_rfc5424(const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t n, const fg::DevTables* t,
                                 uint64_t avg_len, hipStream_t stream, uint64_t* stash, uint32_t stash_blocks, uint32_t strip,
                                 const uint8_t* line_bad, const fg_launch_opts* lo, fg::TicketSlot* tk);
extern "C" uint64_t fg_stash_bytes(uint32_t blocks);
extern "C" int fg_launch_rfc3164(const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t n, const fg::DevTables* t,
                                 const fg::r3164::Cfg* cfg, uint32_t tile_cap, hipStream_t stream, uint32_t strip,
                                 const uint8_t* line_bad, uint8_t* scratch, int regroup);
extern "C" uint64_t fg_rfc3164_scratch_bytes(uint64_t n);
extern "C" uint64_t fg_rfc3164_regroup_from(void);   // lines from which the library regroups by itself  // the regrouped form's scratch: shape keys, block counts, the line permutation
extern "C" int fg_launch_encode_sizes(const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t n, const fg::DevTables* t,
                                      const fg::EncCfg* cfg, uint32_t tile_cap, uint32_t cfg_lds, uint32_t* d_sizes,
                                      uint64_t* d_block_sums, uint8_t* d_status, uint64_t* d_out_offsets, hipStream_t stream);
extern "C" int fg_launch_encode_count(const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t n, const fg::DevTables* t,
                                      const fg::EncCfg* cfg, uint32_t tile_cap, uint32_t cfg_lds, uint32_t* d_sizes,
                                      uint64_t* d_block_sums, uint8_t* d_status, hipStream_t stream);
extern "C" int fg_launch_encode_scan(const uint32_t* d_sizes, uint64_t* d_block_sums, uint64_t n, uint64_t* d_out_offsets, uint64_t base,
                                     hipStream_t stream);
extern "C" int fg_launch_encode_write(const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t n, const fg::DevTables* t,
                                      const fg::EncCfg* cfg, uint32_t tile_cap, uint32_t cfg_lds, const uint64_t* d_out_offsets,
                                      uint8_t* d_out, const uint32_t* d_sizes, hipStream_t stream);
extern "C" uint64_t fg_frame_scratch_bytes(uint64_t nbytes);
extern "C" int fg_launch_frame(const uint8_t* d_bytes, uint64_t nbytes, uint32_t delim, uint8_t* scratch, uint64_t* d_offsets,
                               uint8_t* d_bad, uint64_t cap, uint64_t** d_total_out, hipStream_t stream, int classic);
constexpr uint64_t FG_FRAME_ABORTED = ~0ull;  // *d_total_out after a one-pass launch whose look-back gave up: launch again, classic
extern "C" uint64_t fg_frame_block_bytes(void);
extern "C" uint64_t fg_frame_slice_align(void);
extern "C" int fg_launch_frame_slice(const uint8_t* d_bytes, uint64_t nbytes, uint32_t delim, uint8_t* scratch, uint64_t* d_offsets,
                                     uint8_t* d_bad, uint64_t cap, uint64_t blk0, uint64_t blk1, uint64_t** d_total_out,
                                     hipStream_t stream, const uint8_t* src, int classic);
extern "C" int fg_launch_ltsv(const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t n, const fg::DevTables* t,
                              const fg::LtsvDevCfg* cfg, uint64_t avg_len, hipStream_t stream, uint64_t* stash,
                              uint32_t stash_blocks, uint32_t strip, const uint8_t* line_bad, const fg_launch_opts* lo, fg::TicketSlot* tk);
extern "C" int fg_launch_gelf_general(const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t n, const fg::DevTables* t, hipStream_t stream,
                                      uint32_t strip, const uint8_t* line_bad);
extern "C" int fg_launch_gelf(const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t n, const fg::DevTables* t,
                              uint64_t avg_len, hipStream_t stream, uint64_t* stash, uint32_t stash_blocks, uint32_t strip,
                              const uint8_t* line_bad, const fg_launch_opts* lo, fg::TicketSlot* tk);

extern "C" int fg_launch_gelf_general_dev(const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t n, const fg::DevTables* t, hipStream_t stream,
                                          uint32_t strip, const uint8_t* line_bad, const unsigned long long* n_dev);
// the fused launches (fg_fused.hpp): frame + decode of a raw stream chunk in ONE kernel; g from fg::fused_geometry, scratch of
// fg::fused_scratch_bytes(nbytes, g->S) bytes; *d_total = the launch's two result words (lines, abort flag), device memory
extern "C" int fg_launch_rfc5424_fused(const uint8_t* d_bytes, uint64_t nbytes, const fg::DevTables* t, const fg::FusedGeom* g, hipStream_t stream,
                                       uint64_t* stash, uint32_t stash_blocks, uint32_t strip, int final_, uint64_t* d_offsets, uint64_t cap,
                                       uint8_t* scratch, const fg_launch_opts* lo, unsigned long long** d_total);
extern "C" int fg_launch_ltsv_fused(const uint8_t* d_bytes, uint64_t nbytes, const fg::DevTables* t, const fg::LtsvDevCfg* cfg, const fg::FusedGeom* g,
                                    hipStream_t stream, uint64_t* stash, uint32_t stash_blocks, uint32_t strip, int final_, uint64_t* d_offsets,
                                    uint64_t cap, uint8_t* scratch, const fg_launch_opts* lo, unsigned long long** d_total);
extern "C" int fg_launch_gelf_fused(const uint8_t* d_bytes, uint64_t nbytes, const fg::DevTables* t, const fg::FusedGeom* g, hipStream_t stream,
                                    uint32_t strip, int final_, uint64_t* d_offsets, uint64_t cap, uint8_t* scratch, const fg_launch_opts* lo,
                                    unsigned long long** d_total);
extern "C" int fg_launch_poke64(const uint64_t* d_src, uint64_t* dst_devview, hipStream_t stream);
extern "C" uint64_t fg_merge_scratch_bytes(uint64_t rows);
extern "C" int fg_launch_merge_device(const fg_tables* parts, uint32_t g, const uint64_t* const* d_index, const fg_tables* out, uint8_t* d_src_part,
                                      uint64_t max_rows, uint8_t* scratch, hipStream_t stream);
extern "C" int fg_launch_calib(int mode, const uint8_t* d_src, uint8_t* d_dst, uint64_t nbytes, uint32_t* d_sink, hipStream_t stream);

struct fg_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;  // the pipelined host paths (created on first use): uploads / second lane
    hipStream_t stream3 = nullptr;  // ... downloads
    std::vector<hipEvent_t> ev_slice;  // ... two events per slice: uploaded, decoded
    hipStream_t s_up = nullptr, s_down = nullptr, s_run = nullptr;  // ... which of the three does what
    hipEvent_t ev_ready = nullptr;
    int last_hip = 0;
    fg_launch_opts lo{};  // launch-geometry overrides (fg_set_launch_opts); all zero = the library's own choices
    uint32_t link_bound_waves = 0;  // host pipelines whose tables lie across the link: waves per CU of the decode grid for the duration of
                                    // a call, where the caller's options leave the choice to the library (fg_host_pipeline.cpp LinkBoundGrid)
    uint32_t table_shares = 0;  // a sliced host path: the launches of the batch that share one entry table (fg::entry_chunk), else 0
    int last_host_path = 0;  // fg_last_host_path: which form the last host-buffer decode call took (FG_PATH_*)
    bool timing = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool ev_valid = false;
    // LTSV configuration (owned copies)
    std::vector<std::string> schema_names;
    std::vector<uint8_t> schema_types;
    std::string suffix[4];
    bool has_suffix[4] = {false, false, false, false};
    uint8_t* d_cfg = nullptr;  // device copy of the LTSV configuration (blob | name_off | types)
    // per-wave scratch where entries (SD pairs / LTSV pairs / GELF extras) are parked between the
    // parse and the copy into the entry table (allocated on the first decode; 8 waves on every CU)
    uint64_t* d_stash = nullptr;
    uint64_t* d_stash2 = nullptr;  // the second lane's (fg_transcode_batch)
    uint32_t stash_blocks = 0;
    uint32_t* d_pending = nullptr;  // ring of kPendingRing hand-over words (DevTables::pending), zeroed once
    uint32_t* d_ticket = nullptr;   // ring of ticket counters, one per decode launch (fg::TicketSlot); zeroed once
    std::vector<uint32_t> h_ticket; // ... what each word holds once the launches issued so far have run
    uint32_t ticket_seq = 0;        // ... the next launch's slot
    uint32_t epoch = 0;             // launch counter of this ctx
    bool defer_general = false;     // a sliced host path: GELF's exact form runs once, behind the last slice (fg_finish_deferred_general)
    uint32_t batch_epoch = 0;       // ... and the slices share one hand-over word: the epoch of the batch's first GELF launch
    uint8_t* d_merge = nullptr;     // fg_merge_tables_device: the scan's scratch (dense offsets, block sums, part tags)
    uint64_t d_merge_cap = 0;
    uint32_t* d_sink = nullptr;     // fg_calibrate_device: the word the read-only sweep may write
    uint64_t* d_used = nullptr;     // fg_decode_batch, zero-copy form: the entry counter (the tables themselves are pinned host memory)
    uint8_t* d_frame = nullptr;  // fg_frame_device scratch (delimiter / UTF-8 masks, block counts)
    uint64_t d_frame_cap = 0;
    uint8_t* d_r3164 = nullptr;  // FG_RFC3164, lines regrouped by shape: keys, block counts, permutation (fg_rfc3164.hip)
    uint64_t d_r3164_cap = 0;
    uint8_t* d_fused = nullptr;  // fg_frame_decode_device: the fused launch's scratch (ticket counters, tile counts, block prefixes)
    uint64_t d_fused_cap = 0;
    uint8_t* d_bad = nullptr;    // fg_frame_decode_batch: per-frame UTF-8 verdicts
    uint64_t d_bad_cap = 0;
    uint64_t* h_off = nullptr;   // fg_frame_decode_batch: pinned host copy of the frame offsets
    uint64_t h_off_cap = 0;
    uint64_t* h_cnt = nullptr;   // ... pinned words the pipelined form reads the slices' frame counts through
    double frames_per_byte = 1.0 / 200.0;  // ... what the last raw chunk held (sizes the next one's tables before its frames are counted)
    // RFC3164 configuration: host copies (for fg_clone) + one device block [names | name_off | zone_first | utc_start | utc_off]
    bool r3164_set = false;
    bool r3164_auto_year = false;  // current_year == FG_YEAR_NOW: follow the wall clock like the reference (:179)
    int32_t r3164_year = 1970;
    std::vector<uint8_t*> retired_tz;  // zone blocks replaced at a year change (kernels may still read them; freed at destroy)
    std::vector<std::string> tz_names;
    std::vector<uint32_t> tz_first;
    std::vector<int64_t> tz_start;
    std::vector<int32_t> tz_off;
    uint8_t* d_tz = nullptr;
    fg::r3164::Cfg r3164{};
    uint8_t* d_enc = nullptr;    // fg_encode_gelf_device: static key list + blob, then the per-line sizes
    uint64_t d_enc_cap = 0;
    // fg_encode_device_async: pinned ring the encoder configuration is uploaded from without a host sync
    static constexpr uint32_t kEncRing = 4, kEncSlot = 16 * 1024;
    uint8_t* h_enc_ring = nullptr;
    hipEvent_t ev_enc[kEncRing] = {nullptr, nullptr, nullptr, nullptr};
    uint32_t enc_ring_next = 0;
    fg::LtsvDevCfg ltsv{};
    // staging for fg_decode_batch
    uint8_t* d_bytes = nullptr;
    uint64_t d_bytes_cap = 0;
    uint64_t* d_offsets = nullptr;
    uint64_t d_offsets_cap = 0;
    uint8_t* d_tab = nullptr;  // one device allocation carved into the table arrays
    uint64_t d_tab_cap = 0;
    uint8_t* h_tab = nullptr;  // pinned host mirror
    uint64_t h_tab_cap = 0;
    // fg_transcode_batch: device output (messages | out_offsets | enc_status) and its pinned host mirror
    uint8_t* d_tout = nullptr;
    uint64_t d_tout_cap = 0;
    uint8_t* d_tmeta = nullptr;  // out_offsets[n + 1] then enc_status[n]
    uint64_t d_tmeta_cap = 0;
    uint8_t* h_tout = nullptr;   // pinned: messages | out_offsets | meta | enc_status
    uint64_t h_tout_cap = 0;
};

namespace {

struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) (void)hipSetDevice(dev);
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

#define FG_HIP(ctx, call)                         \
    do {                                          \
        hipError_t e_ = (call);                   \
        if (e_ != hipSuccess) {                   \
            (ctx)->last_hip = (int)e_;            \
            return FG_ERR_HIP;                    \
        }                                         \
    } while (0)

inline uint64_t up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }

// carve `base` into the arrays of an fg_tables (256-byte aligned pieces)
void carve(uint8_t* base, uint64_t n, uint64_t ent_cap, fg_tables* t, uint64_t* total) {
    uint64_t sizes[FG_TABLE_ARRAYS];
    fg_tables_layout(n, ent_cap, sizes);
    uint64_t off = 0;
    uint8_t* p[FG_TABLE_ARRAYS];
    for (int k = 0; k < FG_TABLE_ARRAYS; ++k) {
        p[k] = base ? base + off : nullptr;
        off += up(sizes[k], 256);
    }
    if (total) *total = off;
    if (!t) return;
    t->n = n;
    t->ent_cap = ent_cap;
    t->meta = (uint32_t*)p[0];
    t->ts = (double*)p[1];
    t->hostname = (fg_span*)p[2];
    t->appname = (fg_span*)p[3];
    t->procid = (fg_span*)p[4];
    t->msgid = (fg_span*)p[5];
    t->msg = (fg_span*)p[6];
    t->full_msg = (fg_span*)p[7];
    t->ent_first = (uint32_t*)p[8];
    t->ent_count = (uint32_t*)p[9];
    t->ent_name = (fg_span*)p[10];
    t->ent_val = (uint64_t*)p[11];
    t->ent_type = (uint8_t*)p[12];
    t->ent_flags = (uint8_t*)p[13];
    t->ent_used = (uint64_t*)p[14];
}

fg::DevTables to_dev(const fg_tables& t) {
    fg::DevTables d;
    d.n = t.n;
    d.ent_cap = t.ent_cap;
    d.meta = t.meta;
    d.ts = t.ts;
    d.span[0] = t.hostname;
    d.span[1] = t.appname;
    d.span[2] = t.procid;
    d.span[3] = t.msgid;
    d.span[4] = t.msg;
    d.span[5] = t.full_msg;
    d.ent_first = t.ent_first;
    d.ent_count = t.ent_count;
    d.ent_name = t.ent_name;
    d.ent_val = t.ent_val;
    d.ent_type = t.ent_type;
    d.ent_flags = t.ent_flags;
    d.ent_used = (unsigned long long*)t.ent_used;
    d.pending = nullptr;
    d.epoch = 0;
    d.alloc_chunk = 0;
    d.shares = 1;
    return d;
}

}  // namespace
#include "fg_tile_cap.hpp"
namespace {

int grow_dev(fg_ctx* ctx, void** p, uint64_t* cap, uint64_t need) {
    if (need <= *cap) return FG_OK;
    if (*p) FG_HIP(ctx, hipFree(*p));
    *p = nullptr;
    *cap = 0;
    uint64_t want = up(need + need / 4, 1 << 20);
    FG_HIP(ctx, hipMalloc(p, want));
    *cap = want;
    return FG_OK;
}

int grow_pinned(fg_ctx* ctx, void** p, uint64_t* cap, uint64_t need) {
    if (need <= *cap) return FG_OK;
    if (*p) FG_HIP(ctx, hipHostFree(*p));
    *p = nullptr;
    *cap = 0;
    uint64_t want = up(need + need / 4, 1 << 20);
    FG_HIP(ctx, hipHostMalloc(p, want, hipHostMallocDefault));
    *cap = want;
    return FG_OK;
}

}  // namespace

// one decode launch on `stream` (fg_capi.cpp).  reset_counter = false: a further slice of a batch whose entry counter is already live;
// span_bytes = the bytes the n lines cover (launch geometry is planned from the average line); lane = 1: a second launch that may be in
// flight at the same time (its own entry stash)
extern "C" int fg_finish_deferred_general(fg_ctx* ctx, fg_framing framing, const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t n,
                               const uint8_t* d_bad_utf8, const fg_tables* tables, void* stream);
// one fused launch on `stream` (fg_capi.cpp): *d_total = its result words in ctx scratch.  FG_ERR_UNSUPPORTED: nothing was launched
extern "C" int fg_frame_decode_impl(fg_ctx* ctx, fg_format fmt, fg_framing framing, const uint8_t* d_bytes, uint64_t nbytes, int final,
                                    uint64_t* d_offsets, uint64_t cap, const fg_tables* tables, uint64_t avg_line, void* stream,
                                    unsigned long long** d_total);
extern "C" int fg_decode_frames_impl(fg_ctx* ctx, fg_format fmt, fg_framing framing, const uint8_t* d_bytes, uint64_t nbytes,
                          const uint64_t* d_offsets, uint64_t n, const uint8_t* d_bad_utf8, const fg_tables* tables,
                          void* stream, bool reset_counter, uint64_t span_bytes, uint32_t lane = 0);
