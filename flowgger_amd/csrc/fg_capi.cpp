// fg_capi.cpp -- host side of the C ABI declared in include/fg_hip.h.
//
// Owns: device selection (gfx950 only, no CPU fallback), the per-ctx HIP stream, staging
// buffers for the host-buffer entry point, HIP-event timing of the decode kernels.
// It mirrors what XDecoder::new(&Config) + Box<dyn Decoder+Send>::clone do in the reference
// (src/flowgger/mod.rs:413-422, src/flowgger/decoder/mod.rs:23-36): a ctx is built once from the
// configuration and cloned per connection thread.
#include "fg_ctx.hpp"

namespace {

const char* const kErr5424[] = {
    "",
    "Unsupported BOM",
    "The priority should be inside brackets",
    "Invalid priority",
    "Missing version",
    "Unsupported version",
    "Missing timestamp",
    "Unable to parse the date from RFC3339 to Unix time in RFC5424 decoder",
    "Missing hostname",
    "Missing application name",
    "Missing process id",
    "Missing message id",
    "Missing message data",
    "Missing log message",
    "Malformated RFC5424 message",
    "Missing structured data",
    "Format error in the structured data",
    "Missing ] after structured data",
};
const char* const kErrLtsv[] = {
    "",
    "Invalid severity level",
    "Severity level should be <= 7",
    "Type error; boolean was expected",
    "Type error; f64 was expected",
    "Type error; i64 was expected",
    "Type error; u64 was expected",
    "Missing timestamp",
    "Missing hostname",
    "Unable to parse the English to Unix timestamp in LTSV decoder",
};
const char* const kErrGelf[] = {
    "",
    "Invalid GELF input, unable to parse as a JSON object",
    "Empty GELF input",
    "Invalid GELF timestamp",
    "GELF host name must be a string",
    "GELF short message must be a string",
    "GELF full message must be a string",
    "GELF version must be a string",
    "Unsupported GELF version",
    "Invalid severity level",
    "Invalid severity level (too high)",
    "Invalid value type in structured data",
    "Missing hostname",
};

// Upload the LTSV schema / suffixes once per ctx (LTSVDecoder::new, ltsv_decoder.rs:24-84).
int upload_ltsv_cfg(fg_ctx* ctx) {
    std::vector<uint8_t> blob;
    std::vector<uint32_t> off;
    for (const auto& nme : ctx->schema_names) {
        off.push_back((uint32_t)blob.size());
        blob.insert(blob.end(), nme.begin(), nme.end());
    }
    off.push_back((uint32_t)blob.size());
    fg::LtsvDevCfg c{};
    c.n_schema = (uint32_t)ctx->schema_names.size();
    for (int k = 0; k < 4; ++k) {
        c.suf_off[k] = (uint32_t)blob.size();
        c.suf_len[k] = (uint32_t)ctx->suffix[k].size();
        c.has_suf[k] = ctx->has_suffix[k] ? 1u : 0u;
        blob.insert(blob.end(), ctx->suffix[k].begin(), ctx->suffix[k].end());
    }
    const uint64_t blob_sz = up(blob.size() + 1, 16), off_sz = up(off.size() * 4, 16), ty_sz = up(ctx->schema_types.size() + 1, 16);
    FG_HIP(ctx, hipMalloc((void**)&ctx->d_cfg, blob_sz + off_sz + ty_sz));
    std::vector<uint8_t> host(blob_sz + off_sz + ty_sz, 0);
    if (!blob.empty()) memcpy(host.data(), blob.data(), blob.size());
    memcpy(host.data() + blob_sz, off.data(), off.size() * 4);
    if (!ctx->schema_types.empty()) memcpy(host.data() + blob_sz + off_sz, ctx->schema_types.data(), ctx->schema_types.size());
    FG_HIP(ctx, hipMemcpy(ctx->d_cfg, host.data(), host.size(), hipMemcpyHostToDevice));
    c.blob = ctx->d_cfg;
    c.name_off = (const uint32_t*)(ctx->d_cfg + blob_sz);
    c.types = ctx->d_cfg + blob_sz + off_sz;
    ctx->ltsv = c;
    return FG_OK;
}

}  // namespace

extern "C" {

int fg_abi_version(void) { return FG_ABI_VERSION; }
int fg_last_host_path(const fg_ctx* ctx) { return ctx ? ctx->last_host_path : 0; }

int fg_tables_layout(uint64_t n, uint64_t ent_cap, uint64_t sizes[FG_TABLE_ARRAYS]) {
    if (!sizes) return FG_ERR_ARG;
    sizes[0] = n * 4;
    sizes[1] = n * 8;
    for (int k = 2; k < 8; ++k) sizes[k] = n * 8;
    sizes[8] = n * 4;
    sizes[9] = n * 4;
    sizes[10] = ent_cap * 8;
    sizes[11] = ent_cap * 8;
    sizes[12] = ent_cap;
    sizes[13] = ent_cap;
    sizes[14] = 8;
    return FG_OK;
}

int fg_create(int device, const fg_cfg* cfg, fg_ctx** out) {
    if (!out) return FG_ERR_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return FG_ERR_NO_DEVICE;
    if (device < 0 || device >= count) return FG_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return FG_ERR_NO_DEVICE;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return FG_ERR_NO_DEVICE;  // kernels are gfx950-only
    if (cfg && cfg->n_schema) {
        if (!cfg->schema_names || !cfg->schema_types) return FG_ERR_ARG;
        for (uint32_t i = 0; i < cfg->n_schema; ++i)
            if (!cfg->schema_names[i] || cfg->schema_types[i] > FG_T_U64) return FG_ERR_ARG;
    }
    fg_ctx* ctx = new (std::nothrow) fg_ctx();
    if (!ctx) return FG_ERR_NOMEM;
    ctx->device = device;
    if (cfg) {
        for (uint32_t i = 0; i < cfg->n_schema; ++i) {
            ctx->schema_names.emplace_back(cfg->schema_names[i]);
            ctx->schema_types.push_back(cfg->schema_types[i]);
        }
        const char* s[4] = {cfg->suffix_bool, cfg->suffix_f64, cfg->suffix_i64, cfg->suffix_u64};
        for (int k = 0; k < 4; ++k)
            if (s[k]) {
                ctx->suffix[k] = s[k];
                ctx->has_suffix[k] = true;
            }
    }
    DeviceGuard g(device);
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return FG_ERR_HIP;
    }
    if (upload_ltsv_cfg(ctx) != FG_OK) {
        fg_destroy(ctx);
        return FG_ERR_HIP;
    }
    *out = ctx;
    return FG_OK;
}

static int upload_tz(fg_ctx* ctx);
static int utc_year_now() {
    const time_t t = time(nullptr);
    struct tm g;
    gmtime_r(&t, &g);
    return g.tm_year + 1900;
}
// OffsetDateTime::now_utc().year() is evaluated per parse by the reference (rfc3164_decoder.rs:179); here once per decode
// call when the ctx was configured with FG_YEAR_NOW: a long-lived decoder crosses New Year without a restart.
static int refresh_year(fg_ctx* ctx) {
    if (!ctx->r3164_auto_year) return FG_OK;
    const int y = utc_year_now();
    if (y == ctx->r3164_year) return FG_OK;
    // launches already queued on the caller's streams may still read the old block: it is retired, not freed.  The year is
    // taken over only when the new block is in place -- a failed upload leaves year, block and view as they were, so the next
    // call tries again instead of decoding zone-tagged lines without a zone table.
    const int32_t old_year = ctx->r3164_year;
    uint8_t* const old_tz = ctx->d_tz;
    const fg::r3164::Cfg old_cfg = ctx->r3164;
    ctx->r3164_year = y;
    ctx->d_tz = nullptr;
    const int rc = upload_tz(ctx);
    if (rc != FG_OK) {
        if (ctx->d_tz) (void)hipFree(ctx->d_tz);
        ctx->d_tz = old_tz;
        ctx->r3164 = old_cfg;
        ctx->r3164_year = old_year;
        return rc;
    }
    if (old_tz) ctx->retired_tz.push_back(old_tz);
    return FG_OK;
}

int fg_clone(const fg_ctx* src, fg_ctx** out) {
    if (!src || !out) return FG_ERR_ARG;
    fg_ctx* ctx = new (std::nothrow) fg_ctx();
    if (!ctx) return FG_ERR_NOMEM;
    ctx->device = src->device;
    ctx->lo = src->lo;
    ctx->schema_names = src->schema_names;
    ctx->schema_types = src->schema_types;
    for (int k = 0; k < 4; ++k) {
        ctx->suffix[k] = src->suffix[k];
        ctx->has_suffix[k] = src->has_suffix[k];
    }
    DeviceGuard g(ctx->device);
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return FG_ERR_HIP;
    }
    if (upload_ltsv_cfg(ctx) != FG_OK) {
        fg_destroy(ctx);
        return FG_ERR_HIP;
    }
    if (src->r3164_set) {
        ctx->r3164_set = true;
        ctx->r3164_auto_year = src->r3164_auto_year;
        ctx->r3164_year = src->r3164_year;
        ctx->tz_names = src->tz_names;
        ctx->tz_first = src->tz_first;
        ctx->tz_start = src->tz_start;
        ctx->tz_off = src->tz_off;
        if (upload_tz(ctx) != FG_OK) {
            fg_destroy(ctx);
            return FG_ERR_HIP;
        }
    }
    *out = ctx;
    return FG_OK;
}

void fg_destroy(fg_ctx* ctx) {
    if (!ctx) return;
    DeviceGuard g(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->d_bytes) (void)hipFree(ctx->d_bytes);
    if (ctx->d_offsets) (void)hipFree(ctx->d_offsets);
    if (ctx->d_tab) (void)hipFree(ctx->d_tab);
    if (ctx->d_cfg) (void)hipFree(ctx->d_cfg);
    if (ctx->d_stash) (void)hipFree(ctx->d_stash);
    if (ctx->d_stash2) (void)hipFree(ctx->d_stash2);
    if (ctx->d_pending) (void)hipFree(ctx->d_pending);
    if (ctx->d_ticket) (void)hipFree(ctx->d_ticket);
    if (ctx->d_merge) (void)hipFree(ctx->d_merge);
    if (ctx->d_sink) (void)hipFree(ctx->d_sink);
    if (ctx->d_used) (void)hipFree(ctx->d_used);
    if (ctx->d_frame) (void)hipFree(ctx->d_frame);
    if (ctx->d_fused) (void)hipFree(ctx->d_fused);
    if (ctx->d_r3164) (void)hipFree(ctx->d_r3164);
    if (ctx->d_bad) (void)hipFree(ctx->d_bad);
    if (ctx->d_enc) (void)hipFree(ctx->d_enc);
    if (ctx->h_enc_ring) (void)hipHostFree(ctx->h_enc_ring);
    for (hipEvent_t e : ctx->ev_enc)
        if (e) (void)hipEventDestroy(e);
    if (ctx->d_tz) (void)hipFree(ctx->d_tz);
    for (uint8_t* p : ctx->retired_tz) (void)hipFree(p);
    if (ctx->h_off) (void)hipHostFree(ctx->h_off);
    if (ctx->h_cnt) (void)hipHostFree(ctx->h_cnt);
    if (ctx->h_tab) (void)hipHostFree(ctx->h_tab);
    if (ctx->d_tout) (void)hipFree(ctx->d_tout);
    if (ctx->d_tmeta) (void)hipFree(ctx->d_tmeta);
    if (ctx->h_tout) (void)hipHostFree(ctx->h_tout);
    if (ctx->stream2) (void)hipStreamSynchronize(ctx->stream2);
    if (ctx->stream3) (void)hipStreamSynchronize(ctx->stream3);
    for (hipEvent_t e : ctx->ev_slice) (void)hipEventDestroy(e);
    if (ctx->stream3) (void)hipStreamDestroy(ctx->stream3);
    if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
    if (ctx->ev_ready) (void)hipEventDestroy(ctx->ev_ready);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int fg_last_hip_error(const fg_ctx* ctx) { return ctx ? ctx->last_hip : 0; }

int fg_set_launch_opts(fg_ctx* ctx, const fg_launch_opts* opts) {
    if (!ctx) return FG_ERR_ARG;
    if (opts && (opts->lines_per_group > 64 || (opts->gelf_window_kib && (opts->gelf_window_kib < 2 || opts->gelf_window_kib > 6)))) return FG_ERR_ARG;
    ctx->lo = opts ? *opts : fg_launch_opts{};
    ctx->lo.flags &= ~(uint32_t)FG_LO_RESERVED;
    return FG_OK;
}

int fg_set_timing(fg_ctx* ctx, int enabled) {
    if (!ctx) return FG_ERR_ARG;
    DeviceGuard g(ctx->device);
    if (enabled && !ctx->ev0) {
        FG_HIP(ctx, hipEventCreate(&ctx->ev0));
        FG_HIP(ctx, hipEventCreate(&ctx->ev1));
    }
    ctx->timing = enabled != 0;
    ctx->ev_valid = false;
    return FG_OK;
}

int fg_ticket_ring_check(fg_ctx* ctx) {
    if (!ctx) return FG_ERR_ARG;
    DeviceGuard g(ctx->device);
    if (!ctx->d_ticket) return 0;  // no decode launch yet
    FG_HIP(ctx, hipDeviceSynchronize());  // (launches may have gone to any stream the caller passed)
    std::vector<uint32_t> dev(ctx->h_ticket.size());
    FG_HIP(ctx, hipMemcpy(dev.data(), ctx->d_ticket, dev.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    int bad = 0;
    for (size_t k = 0; k < dev.size(); ++k) bad += dev[k] != ctx->h_ticket[k] ? 1 : 0;
    return bad;
}

int fg_last_kernel_ms(fg_ctx* ctx, float* ms) {
    if (!ctx || !ms) return FG_ERR_ARG;
    *ms = 0.f;
    if (!ctx->timing || !ctx->ev_valid) return FG_OK;
    DeviceGuard g(ctx->device);
    FG_HIP(ctx, hipEventSynchronize(ctx->ev1));
    FG_HIP(ctx, hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
    return FG_OK;
}

int fg_calibrate_device(fg_ctx* ctx, int mode, const uint8_t* d_src, uint8_t* d_dst, uint64_t nbytes, void* stream) {
    if (!ctx || !d_src || mode < FG_CALIB_COPY || mode > FG_CALIB_COPY_FLAT || (mode != FG_CALIB_READ && !d_dst)) return FG_ERR_ARG;
    if ((((uintptr_t)d_src) | ((uintptr_t)d_dst)) & 15u) return FG_ERR_ARG;
    DeviceGuard g(ctx->device);
    if (!ctx->d_sink) FG_HIP(ctx, hipMalloc((void**)&ctx->d_sink, 256));
    hipStream_t s = stream == FG_STREAM_OWN ? ctx->stream : (hipStream_t)stream;
    const int rc = fg_launch_calib(mode, d_src, d_dst, nbytes, ctx->d_sink, s);
    if (rc != 0) {
        ctx->last_hip = rc;
        return FG_ERR_HIP;
    }
    return FG_OK;
}

int fg_merge_tables_device(fg_ctx* ctx, const fg_tables* parts, uint32_t g, const uint64_t* const* d_index, const fg_tables* out,
                           uint8_t* d_src_part, void* stream) {
    if (!ctx || !parts || !d_index || !out || g == 0 || g > 8) return FG_ERR_ARG;
    uint64_t rows = 0, cap = 0, max_rows = 0, max_cap = 0;
    // every column the merge kernels read or write must be there (a table without its entry counter was a device fault)
    auto complete = [](const fg_tables& t, bool entries) {
        const bool rows_ok = t.n == 0 || (t.meta && t.ts && t.hostname && t.appname && t.procid && t.msgid && t.msg && t.full_msg && t.ent_first &&
                                          t.ent_count);
        const bool ent_ok = !entries || (t.ent_name && t.ent_val && t.ent_type && t.ent_flags);
        return rows_ok && ent_ok && t.ent_used != nullptr;
    };
    if (!complete(*out, out->ent_cap != 0)) return FG_ERR_ARG;
    for (uint32_t k = 0; k < g; ++k) {
        if (parts[k].n && !d_index[k]) return FG_ERR_ARG;
        if (!complete(parts[k], parts[k].ent_cap != 0)) return FG_ERR_ARG;
        rows += parts[k].n;
        cap += parts[k].ent_cap;
        if (parts[k].n > max_rows) max_rows = parts[k].n;
        if (parts[k].ent_cap > max_cap) max_cap = parts[k].ent_cap;
    }
    if (out->n != rows || out->ent_cap < cap || cap > 0xFFFFFFFFull) return FG_ERR_ARG;
    DeviceGuard gd(ctx->device);
    hipStream_t s = stream == FG_STREAM_OWN ? ctx->stream : (hipStream_t)stream;
    (void)max_cap;
    int grc;
    if ((grc = grow_dev(ctx, (void**)&ctx->d_merge, &ctx->d_merge_cap, fg_merge_scratch_bytes(rows))) != FG_OK) return grc;
    const int rc = fg_launch_merge_device(parts, g, d_index, out, d_src_part, max_rows, ctx->d_merge, s);
    if (rc != 0) {
        ctx->last_hip = rc;
        return FG_ERR_HIP;
    }
    return FG_OK;
}

int fg_decode_batch_device(fg_ctx* ctx, fg_format fmt, const uint8_t* d_bytes, uint64_t nbytes,
                           const uint64_t* d_offsets, uint64_t n, const fg_tables* tables, void* stream) {
    return fg_decode_frames_device(ctx, fmt, FG_FRAME_NONE, d_bytes, nbytes, d_offsets, n, nullptr, tables, stream);
}

int fg_frame_device(fg_ctx* ctx, fg_framing framing, const uint8_t* d_bytes, uint64_t nbytes, uint64_t* d_offsets,
                    uint8_t* d_bad_utf8, uint64_t cap_frames, uint64_t* n_frames, void* stream) {
    if (!ctx || !d_offsets || !d_bad_utf8 || !n_frames || (nbytes && !d_bytes)) return FG_ERR_ARG;
    if (framing != FG_FRAME_LINE && framing != FG_FRAME_NUL) return FG_ERR_UNSUPPORTED;
    if (((uintptr_t)d_bytes & 15u) != 0) return FG_ERR_ARG;
    DeviceGuard g(ctx->device);
    hipStream_t s = stream == FG_STREAM_OWN ? ctx->stream : (hipStream_t)stream;
    *n_frames = 0;
    if (nbytes == 0) return FG_OK;
    int rc;
    if ((rc = grow_dev(ctx, (void**)&ctx->d_frame, &ctx->d_frame_cap, fg_frame_scratch_bytes(nbytes))) != FG_OK) return rc;
    uint64_t total = 0;
    for (int classic = (ctx->lo.flags & FG_LO_FRAME_CLASSIC) ? 1 : (ctx->lo.flags & FG_LO_FRAME_SELFTEST_STALL) ? 2 : 0;; classic = 1) {
        uint64_t* d_total = nullptr;
        int lrc = fg_launch_frame(d_bytes, nbytes, framing == FG_FRAME_LINE ? 0x0Au : 0x00u, ctx->d_frame, d_offsets, d_bad_utf8,
                                  cap_frames, &d_total, s, classic);
        if (lrc != 0) {
            ctx->last_hip = lrc;
            return FG_ERR_HIP;
        }
        // frames = delimiters (+1 when the stream does not end with one)
        FG_HIP(ctx, hipMemcpyAsync(&total, d_total, 8, hipMemcpyDeviceToHost, s));
        FG_HIP(ctx, hipStreamSynchronize(s));
        if (total != FG_FRAME_ABORTED || classic == 1) break;  // (the one-pass scan gave up waiting on a tile: the three-kernel form)
    }
    if (total + 1 > cap_frames) {
        *n_frames = total + 1;
        return FG_ERR_ENT_OVERFLOW;
    }
    uint64_t last_end = 0;
    FG_HIP(ctx, hipMemcpyAsync(&last_end, d_offsets + total, 8, hipMemcpyDeviceToHost, s));
    FG_HIP(ctx, hipStreamSynchronize(s));
    *n_frames = last_end == nbytes ? total : total + 1;
    return FG_OK;
}

int fg_decode_frames_device(fg_ctx* ctx, fg_format fmt, fg_framing framing, const uint8_t* d_bytes, uint64_t nbytes,
                            const uint64_t* d_offsets, uint64_t n, const uint8_t* d_bad_utf8, const fg_tables* tables,
                            void* stream) {
    return fg_decode_frames_impl(ctx, fmt, framing, d_bytes, nbytes, d_offsets, n, d_bad_utf8, tables, stream, true, nbytes);
}

// reset_counter = false: a further slice of a batch whose entry counter is already live (the
// pipelined host path decodes one batch as several slices on two streams).  span_bytes = the bytes
// the n lines cover (the launch geometry is planned from the average line length; nbytes is only
// the readable range of d_bytes and, for a slice, covers the whole batch).
int fg_decode_frames_impl(fg_ctx* ctx, fg_format fmt, fg_framing framing, const uint8_t* d_bytes, uint64_t nbytes,
                              const uint64_t* d_offsets, uint64_t n, const uint8_t* d_bad_utf8, const fg_tables* tables,
                              void* stream, bool reset_counter, uint64_t span_bytes, uint32_t lane) {
    if ((int)framing < 0 || (int)framing > 2) return FG_ERR_ARG;
    if (!ctx || !tables || (n && (!d_offsets || !tables->meta))) return FG_ERR_ARG;
    if (nbytes && !d_bytes) return FG_ERR_ARG;
    if (((uintptr_t)d_bytes & 15u) != 0) return FG_ERR_ARG;
    if (tables->n < n) return FG_ERR_ARG;
    if (tables->ent_cap > 0xFFFFFFFFull) return FG_ERR_ARG;
    DeviceGuard g(ctx->device);
    hipStream_t s = stream == FG_STREAM_OWN ? ctx->stream : (hipStream_t)stream;
    fg::DevTables dt = to_dev(*tables);
    if (dt.ent_used && reset_counter) FG_HIP(ctx, hipMemsetAsync(dt.ent_used, 0, 8, s));
    if (n == 0) return FG_OK;
    if (!ctx->d_stash) {
        hipDeviceProp_t prop;
        FG_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
        uint32_t blocks = 8u * (uint32_t)prop.multiProcessorCount;
        FG_HIP(ctx, hipMalloc((void**)&ctx->d_stash, fg_stash_bytes(blocks)));
        ctx->stash_blocks = blocks;
    }
    uint64_t* stash = ctx->d_stash;
    if (lane) {  // a second launch that may be in flight at the same time (fg_transcode_batch's second lane)
        if (!ctx->d_stash2) FG_HIP(ctx, hipMalloc((void**)&ctx->d_stash2, fg_stash_bytes(ctx->stash_blocks)));
        stash = ctx->d_stash2;
    }
    constexpr uint32_t kPendingRing = 1024;
    if (!ctx->d_pending) {
        FG_HIP(ctx, hipMalloc((void**)&ctx->d_pending, kPendingRing * sizeof(uint32_t)));
        FG_HIP(ctx, hipMemset(ctx->d_pending, 0, kPendingRing * sizeof(uint32_t)));
    }
    dt.shares = ctx->table_shares ? ctx->table_shares : 1u;
    fg_launch_opts lo_call = ctx->lo;
    if (ctx->link_bound_waves && lo_call.waves_per_cu == 0) lo_call.waves_per_cu = ctx->link_bound_waves;
    if (ctx->defer_general && fmt == FG_GELF) {
        // the slices of one batch: one hand-over word for all of them, the exact form once behind the last (fg_finish_deferred_general)
        if (!ctx->batch_epoch) {
            ctx->epoch += 1u;
            if (ctx->epoch == 0u) ctx->epoch = 1u;
            ctx->batch_epoch = ctx->epoch;
        }
        dt.epoch = ctx->batch_epoch;
        lo_call.flags |= (uint32_t)FG_LO_RESERVED;
    } else {
        ctx->epoch += 1u;
        if (ctx->epoch == 0u) ctx->epoch = 1u;  // (0 = the ring's initial content: never a valid epoch)
        dt.epoch = ctx->epoch;
    }
    dt.pending = ctx->d_pending + (dt.epoch % kPendingRing);
    // the launch's ticket counter (dynamic chunk dispatch, fg_pipeline.hpp): one word of a ring per LAUNCH -- the slices of a batch may be
    // in flight on two streams at once and must not draw from one counter -- never reset: the host knows what every launch adds
    constexpr uint32_t kTicketRing = 1024;
    if (!ctx->d_ticket) {
        FG_HIP(ctx, hipMalloc((void**)&ctx->d_ticket, kTicketRing * sizeof(uint32_t)));
        FG_HIP(ctx, hipMemset(ctx->d_ticket, 0, kTicketRing * sizeof(uint32_t)));
        ctx->h_ticket.assign(kTicketRing, 0u);
    }
    const uint32_t tslot = ctx->ticket_seq++ % kTicketRing;
    fg::TicketSlot tk_slot{ctx->d_ticket + tslot, &ctx->h_ticket[tslot]};
    fg::TicketSlot* const tk = (ctx->lo.flags & FG_LO_STATIC_CHUNKS) ? nullptr : &tk_slot;
    if (ctx->timing) FG_HIP(ctx, hipEventRecord(ctx->ev0, s));
    int rc;
    const uint64_t avg_len = (span_bytes + n - 1) / n;
    switch (fmt) {
        case FG_RFC5424:
            rc = fg_launch_rfc5424(d_bytes, d_offsets, n, &dt, avg_len, s, stash, ctx->stash_blocks, (uint32_t)framing,
                                   d_bad_utf8, &lo_call, tk);
            break;
        case FG_LTSV:
            rc = fg_launch_ltsv(d_bytes, d_offsets, n, &dt, &ctx->ltsv, avg_len, s, stash, ctx->stash_blocks,
                                (uint32_t)framing, d_bad_utf8, &lo_call, tk);
            break;
        case FG_GELF:
            rc = fg_launch_gelf(d_bytes, d_offsets, n, &dt, avg_len, s, stash, ctx->stash_blocks,
                                (uint32_t)framing, d_bad_utf8, &lo_call, tk);
            break;
        case FG_RFC3164:
            if (!ctx->r3164_set) return FG_ERR_ARG;  // fg_set_rfc3164 first
            if ((rc = refresh_year(ctx)) != FG_OK) return rc;
{
                // (large batches: lines regrouped by shape -- fg_rfc3164.hip; the scratch is 16 bytes per line: the slow shapes' lists)
                const int regroup = (ctx->lo.flags & FG_LO_RFC3164_NO_REGROUP) ? 2 : (ctx->lo.flags & FG_LO_RFC3164_REGROUP) ? 1 : 0;
                uint8_t* scratch = nullptr;
                if (regroup != 2 && n < 0xFFFFFFFFull && (regroup == 1 || n >= fg_rfc3164_regroup_from()) && lane == 0u) {
                    if ((rc = grow_dev(ctx, (void**)&ctx->d_r3164, &ctx->d_r3164_cap, fg_rfc3164_scratch_bytes(n))) != FG_OK) return rc;
                    scratch = ctx->d_r3164;
                }
                rc = fg_launch_rfc3164(d_bytes, d_offsets, n, &dt, &ctx->r3164, pick_tile_cap(ctx, span_bytes, n, 56 * 1024), s, (uint32_t)framing,
                                       d_bad_utf8, scratch, regroup);
            }
            break;
        default:
            return FG_ERR_UNSUPPORTED;
    }
    if (rc != 0) {
        // The launcher has already moved the host's copy of the launch's ticket word on (take_tickets) and the kernel that would have drawn
        // those tickets did not run: word and copy would differ for good, and every later launch on this slot of the ring would compute
        // chunk indices from a wrong base -- rows silently left undecoded.  Both go back to zero, behind whatever is queued.
        (void)hipMemsetAsync(ctx->d_ticket + tslot, 0, sizeof(uint32_t), s);
        ctx->h_ticket[tslot] = 0u;
        ctx->last_hip = rc;
        return FG_ERR_HIP;
    }
    if (ctx->timing) {
        FG_HIP(ctx, hipEventRecord(ctx->ev1, s));
        ctx->ev_valid = true;
    }
    return FG_OK;
}


// FRAME + DECODE IN ONE KERNEL (fg_fused.hpp): the raw stream chunk d_bytes[0 .. nbytes) is framed by the decode kernel itself --
// rows into `tables` (tables->n = the rows it holds), frame starts into d_offsets (tables->n + 2 entries), and two result words in the
// ctx's scratch: (*d_total)[0] = the frames of the chunk (beyond tables->n: those rows were not written), (*d_total)[1] != 0 = a wave
// gave up waiting on the look-back (nothing is valid: run the separate framing + decode kernels).  Asynchronous on `stream`; nothing
// visits the host.  avg_line = the average frame length to plan for (0: the ctx's experience).  FG_ERR_UNSUPPORTED = this format /
// line length keeps the separate framing pass (RFC3164; long lines, whose kernels stage heads only): nothing was launched.
int fg_frame_decode_impl(fg_ctx* ctx, fg_format fmt, fg_framing framing, const uint8_t* d_bytes, uint64_t nbytes, int final, uint64_t* d_offsets,
                         uint64_t cap, const fg_tables* tables, uint64_t avg_line, void* stream, unsigned long long** d_total) {
    if (!ctx || !tables || !d_offsets || !d_total || (nbytes && !d_bytes)) return FG_ERR_ARG;
    if (framing != FG_FRAME_LINE && framing != FG_FRAME_NUL) return FG_ERR_ARG;
    if (((uintptr_t)d_bytes & 15u) != 0 || tables->n < cap || (cap && !tables->meta) || tables->ent_cap > 0xFFFFFFFFull) return FG_ERR_ARG;
    if (fmt != FG_RFC5424 && fmt != FG_LTSV && fmt != FG_GELF) return FG_ERR_UNSUPPORTED;
    if (nbytes == 0 || cap == 0) return FG_ERR_UNSUPPORTED;
    if (avg_line == 0) avg_line = ctx->frames_per_byte > 0.0 ? (uint64_t)(1.0 / ctx->frames_per_byte + 0.5) : 200u;
    fg_launch_opts lo_call = ctx->lo;
    // Across the link the queue of outstanding reads IS the latency of every dependent one (a tile staged on, a tail scan, the exact
    // GELF form): 2048 waves x a 15 KiB window are 30 MB = half a millisecond at 57 GB/s, and a wave that waits that long holds the
    // look-back of every tile behind it.  Two waves per CU keep the link as full (a few hundred KB in flight do) with a sixteenth of
    // the queue: cfg2 0.864 -> 0.900 of the link, structured data 0.870 -> 0.878 (profiles/r06y_fused_host.log); GELF needs the
    // waves for its arithmetic (one per CU: 0.63) and keeps the link-bound grid of the other zero-copy launches.
    if (ctx->link_bound_waves && lo_call.waves_per_cu == 0) lo_call.waves_per_cu = fmt == FG_GELF ? ctx->link_bound_waves : 2u;
    const fg::FusedGeom g = fg::fused_geometry(fmt, avg_line, lo_call, ctx->link_bound_waves != 0);
    if (!g.ok) return FG_ERR_UNSUPPORTED;
    DeviceGuard guard(ctx->device);
    hipStream_t s = stream == FG_STREAM_OWN ? ctx->stream : (hipStream_t)stream;
    int rc;
    if ((rc = grow_dev(ctx, (void**)&ctx->d_fused, &ctx->d_fused_cap, fg::fused_scratch_bytes(nbytes, g.S))) != FG_OK) return rc;
    fg::DevTables dt = to_dev(*tables);
    if (dt.ent_used) FG_HIP(ctx, hipMemsetAsync(dt.ent_used, 0, 8, s));
    if (!ctx->d_stash) {
        hipDeviceProp_t prop;
        FG_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
        uint32_t blocks = 8u * (uint32_t)prop.multiProcessorCount;
        FG_HIP(ctx, hipMalloc((void**)&ctx->d_stash, fg_stash_bytes(blocks)));
        ctx->stash_blocks = blocks;
    }
    constexpr uint32_t kPendingRing = 1024;
    if (!ctx->d_pending) {
        FG_HIP(ctx, hipMalloc((void**)&ctx->d_pending, kPendingRing * sizeof(uint32_t)));
        FG_HIP(ctx, hipMemset(ctx->d_pending, 0, kPendingRing * sizeof(uint32_t)));
    }
    ctx->epoch += 1u;
    if (ctx->epoch == 0u) ctx->epoch = 1u;
    dt.epoch = ctx->epoch;
    dt.pending = ctx->d_pending + (dt.epoch % kPendingRing);
    dt.n = cap;
    if (ctx->timing) FG_HIP(ctx, hipEventRecord(ctx->ev0, s));
    int lrc;
    switch (fmt) {
        case FG_RFC5424:
            lrc = fg_launch_rfc5424_fused(d_bytes, nbytes, &dt, &g, s, ctx->d_stash, ctx->stash_blocks, (uint32_t)framing, final, d_offsets, cap, ctx->d_fused,
                                          &lo_call, d_total);
            break;
        case FG_LTSV:
            lrc = fg_launch_ltsv_fused(d_bytes, nbytes, &dt, &ctx->ltsv, &g, s, ctx->d_stash, ctx->stash_blocks, (uint32_t)framing, final, d_offsets, cap,
                                       ctx->d_fused, &lo_call, d_total);
            break;
        default:
            lrc = fg_launch_gelf_fused(d_bytes, nbytes, &dt, &g, s, (uint32_t)framing, final, d_offsets, cap, ctx->d_fused, &lo_call, d_total);
            // the lines the fast form handed back: the exact form over the offsets the launch wrote (rows = its frame count, read on the device)
            if (lrc == 0) lrc = fg_launch_gelf_general_dev(d_bytes, d_offsets, cap, &dt, s, (uint32_t)framing, nullptr, *d_total);
            break;
    }
    if (lrc != 0) {
        ctx->last_hip = lrc;
        return FG_ERR_HIP;
    }
    if (ctx->timing) {
        FG_HIP(ctx, hipEventRecord(ctx->ev1, s));
        ctx->ev_valid = true;
    }
    return FG_OK;
}

int fg_frame_decode_device(fg_ctx* ctx, fg_format fmt, fg_framing framing, const uint8_t* d_bytes, uint64_t nbytes, int final, uint64_t* d_offsets,
                           uint64_t cap_frames, const fg_tables* tables, uint64_t avg_line_hint, uint64_t* d_result, void* stream) {
    if (!d_result) return FG_ERR_ARG;
    unsigned long long* d_total = nullptr;
    const int rc = fg_frame_decode_impl(ctx, fmt, framing, d_bytes, nbytes, final, d_offsets, cap_frames, tables, avg_line_hint, stream, &d_total);
    if (rc != FG_OK) return rc;
    DeviceGuard guard(ctx->device);
    hipStream_t s = stream == FG_STREAM_OWN ? ctx->stream : (hipStream_t)stream;
    FG_HIP(ctx, hipMemcpyAsync(d_result, d_total, 16, hipMemcpyDeviceToDevice, s));
    return FG_OK;
}

#if defined(FG_FUSED_STATS)
// measurement variant only (FG_BUILD_VARIANT=stats): the counters / phase clocks the last fused launch of this ctx left (fg_fused.hpp)
extern "C" int fg_debug_fused_stats(fg_ctx* ctx, uint64_t out[16]) {
    if (!ctx || !ctx->d_fused) return FG_ERR_ARG;
    DeviceGuard guard(ctx->device);
    FG_HIP(ctx, hipDeviceSynchronize());
    FG_HIP(ctx, hipMemcpy(out, ctx->d_fused + fg::kFusedCounters * fg::kFusedCounterStride * 4u, 128, hipMemcpyDeviceToHost));
    return FG_OK;
}
#endif

// The exact form of GELF for ALL rows of a batch whose slices were decoded with ctx->defer_general set (a no-op when none was).
int fg_finish_deferred_general(fg_ctx* ctx, fg_framing framing, const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t n,
                               const uint8_t* d_bad_utf8, const fg_tables* tables, void* stream) {
    const uint32_t epoch = ctx->batch_epoch;
    ctx->batch_epoch = 0;
    ctx->defer_general = false;
    if (!epoch || n == 0) return FG_OK;
    DeviceGuard g(ctx->device);
    hipStream_t s = stream == FG_STREAM_OWN ? ctx->stream : (hipStream_t)stream;
    fg::DevTables dt = to_dev(*tables);
    dt.n = n;
    dt.epoch = epoch;
    dt.pending = ctx->d_pending + (epoch % 1024u);
    const int rc = fg_launch_gelf_general(d_bytes, d_offsets, n, &dt, s, (uint32_t)framing, d_bad_utf8);
    if (rc != 0) {
        ctx->last_hip = rc;
        return FG_ERR_HIP;
    }
    return FG_OK;
}

namespace {
// fg_encode_device (total != nullptr: synchronises for the configuration's entry count and for the total) and
// fg_encode_device_async (total == nullptr: nothing on the host waits; ent_hint replaces the entry count read back)
int encode_device_impl(fg_ctx* ctx, fg_format src_fmt, const fg_encode_cfg* ecfg, const uint8_t* d_bytes, uint64_t nbytes,
                       const uint64_t* d_offsets, uint64_t n, const fg_tables* tables, uint8_t* d_out, uint64_t out_cap,
                       uint64_t* d_out_offsets, uint8_t* d_enc_status, uint64_t* total, uint64_t ent_hint, void* stream) {
    const bool async = total == nullptr;
    if (!ctx || !ecfg || !tables || !d_out_offsets || (n && (!d_offsets || !tables->meta))) return FG_ERR_ARG;
    if (async && !d_out) return FG_ERR_ARG;
    if ((int)src_fmt < 0 || (int)src_fmt > (int)FG_RFC3164) return FG_ERR_ARG;
    if (tables->n < n) return FG_ERR_ARG;
    if (((uintptr_t)d_bytes & 15u) != 0) return FG_ERR_ARG;  // (the emitters' global reader loads aligned 16-byte chunks: as the decoders require)
    DeviceGuard g(ctx->device);
    hipStream_t s = stream == FG_STREAM_OWN ? ctx->stream : (hipStream_t)stream;
    if (total) *total = 0;
    fg::EncCfgHost h;
    if (!fg::build_enc_cfg(src_fmt, ecfg, ctx->suffix, ctx->has_suffix, &h)) return FG_ERR_ARG;
    const uint64_t keys_bytes = up(h.keys.size() * sizeof(fg::StaticKey), 16), blob_bytes = up(h.blob.size() + 16, 256);
    const uint64_t cfg_bytes = up(keys_bytes + blob_bytes, 256);
    int rc;
    if ((rc = grow_dev(ctx, (void**)&ctx->d_enc, &ctx->d_enc_cap, cfg_bytes + up(n * 4 + 4, 256) + up((n / 64 + 2) * 8, 256))) != FG_OK) return rc;
    uint64_t ent_used = ~0ull;
    if (async && cfg_bytes <= fg_ctx::kEncSlot) {
        // the configuration through a pinned ring: the copy is queued, nothing waits (a slot is reused four calls later; its
        // event has long fired by then)
        if (!ctx->h_enc_ring) {
            FG_HIP(ctx, hipHostMalloc((void**)&ctx->h_enc_ring, (size_t)fg_ctx::kEncRing * fg_ctx::kEncSlot, hipHostMallocDefault));
            for (uint32_t k = 0; k < fg_ctx::kEncRing; ++k) FG_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_enc[k], hipEventDisableTiming));
        }
        const uint32_t slot = ctx->enc_ring_next++ % fg_ctx::kEncRing;
        FG_HIP(ctx, hipEventSynchronize(ctx->ev_enc[slot]));
        uint8_t* hp = ctx->h_enc_ring + (size_t)slot * fg_ctx::kEncSlot;
        memset(hp, 0, cfg_bytes);
        if (!h.keys.empty()) memcpy(hp, h.keys.data(), h.keys.size() * sizeof(fg::StaticKey));
        if (!h.blob.empty()) memcpy(hp + keys_bytes, h.blob.data(), h.blob.size());
        FG_HIP(ctx, hipMemcpyAsync(ctx->d_enc, hp, cfg_bytes, hipMemcpyHostToDevice, s));
        FG_HIP(ctx, hipEventRecord(ctx->ev_enc[slot], s));
        ent_used = ent_hint;
    } else {
        std::vector<uint8_t> host(cfg_bytes, 0);
        if (!h.keys.empty()) memcpy(host.data(), h.keys.data(), h.keys.size() * sizeof(fg::StaticKey));
        if (!h.blob.empty()) memcpy(host.data() + keys_bytes, h.blob.data(), h.blob.size());
        // (synchronous copy of a few hundred bytes: `host` goes out of scope at return); the same sync brings back the
        // number of entries the decode produced, which sizes the GELF ranking scratch
        FG_HIP(ctx, hipMemcpyAsync(ctx->d_enc, host.data(), host.size(), hipMemcpyHostToDevice, s));
        if (!async && ecfg->encoder == FG_ENC_GELF && tables->ent_used)
            FG_HIP(ctx, hipMemcpyAsync(&ent_used, tables->ent_used, 8, hipMemcpyDeviceToHost, s));
        FG_HIP(ctx, hipStreamSynchronize(s));
        if (async) ent_used = ent_hint;
    }
    fg::EncCfg cfg = h.cfg;
    cfg.keys = reinterpret_cast<const fg::StaticKey*>(ctx->d_enc);
    cfg.blob = ctx->d_enc + keys_bytes;
    if (ent_used == 0) cfg.sort_slots = 1;           // no pairs at all (e.g. RFC5424 without structured data)
    else if (ent_used <= 2 * n) cfg.sort_slots = 8;  // on average <= 2 pairs per line
    // mirror [static keys | blob] in LDS when it is small (it nearly always is)
    const uint32_t cfg_lds = keys_bytes + h.blob.size() <= 4096 ? (uint32_t)up(keys_bytes + h.blob.size(), 16) : 0u;
    uint32_t* d_sizes = reinterpret_cast<uint32_t*>(ctx->d_enc + cfg_bytes);
    uint64_t* d_block_sums = reinterpret_cast<uint64_t*>(ctx->d_enc + cfg_bytes + up(n * 4 + 4, 256));
    fg::DevTables dt = to_dev(*tables);
    if (n == 0) {
        FG_HIP(ctx, hipMemsetAsync(d_out_offsets, 0, 8, s));
        return FG_OK;
    }
    // LDS tile of a 64-line group: average group + 6.25 % + 512 B, 4..40 KiB (longer groups read from global memory);
    // the tile is what limits the waves per CU (the emitters are latency-bound: occupancy is throughput)
    const uint32_t tile_cap = pick_tile_cap(ctx, nbytes, n, 40 * 1024, 1);
    if (ctx->timing) FG_HIP(ctx, hipEventRecord(ctx->ev0, s));
    int lrc = fg_launch_encode_sizes(d_bytes, d_offsets, n, &dt, &cfg, tile_cap, cfg_lds, d_sizes, d_block_sums, d_enc_status, d_out_offsets, s);
    if (lrc != 0) {
        ctx->last_hip = lrc;
        return FG_ERR_HIP;
    }
    if (!async) {
        FG_HIP(ctx, hipMemcpyAsync(total, d_out_offsets + n, 8, hipMemcpyDeviceToHost, s));
        FG_HIP(ctx, hipStreamSynchronize(s));
        if (!d_out) return FG_OK;  // sizing call
        if (*total > out_cap) return FG_ERR_ENT_OVERFLOW;
    } else {
        cfg.out_cap = out_cap ? out_cap : 1;  // the write kernel reads out_offsets[n] itself and leaves d_out alone when it exceeds this
    }
    lrc = fg_launch_encode_write(d_bytes, d_offsets, n, &dt, &cfg, tile_cap, cfg_lds, d_out_offsets, d_out, d_sizes, s);
    if (lrc != 0) {
        ctx->last_hip = lrc;
        return FG_ERR_HIP;
    }
    if (ctx->timing) {  // count + scan + write (fg_last_kernel_ms)
        FG_HIP(ctx, hipEventRecord(ctx->ev1, s));
        ctx->ev_valid = true;
    }
    return FG_OK;
}
}  // namespace

int fg_encode_device(fg_ctx* ctx, fg_format src_fmt, const fg_encode_cfg* ecfg, const uint8_t* d_bytes, uint64_t nbytes,
                     const uint64_t* d_offsets, uint64_t n, const fg_tables* tables, uint8_t* d_out, uint64_t out_cap,
                     uint64_t* d_out_offsets, uint8_t* d_enc_status, uint64_t* total, void* stream) {
    if (!total) return FG_ERR_ARG;
    return encode_device_impl(ctx, src_fmt, ecfg, d_bytes, nbytes, d_offsets, n, tables, d_out, out_cap, d_out_offsets, d_enc_status, total, 0, stream);
}

int fg_encode_device_async(fg_ctx* ctx, fg_format src_fmt, const fg_encode_cfg* ecfg, const uint8_t* d_bytes, uint64_t nbytes,
                           const uint64_t* d_offsets, uint64_t n, const fg_tables* tables, uint8_t* d_out, uint64_t out_cap,
                           uint64_t* d_out_offsets, uint8_t* d_enc_status, uint64_t ent_hint, void* stream) {
    return encode_device_impl(ctx, src_fmt, ecfg, d_bytes, nbytes, d_offsets, n, tables, d_out, out_cap, d_out_offsets, d_enc_status, nullptr,
                              ent_hint, stream);
}

int fg_encode_gelf_device(fg_ctx* ctx, fg_format src_fmt, const uint8_t* d_bytes, uint64_t nbytes, const uint64_t* d_offsets,
                          uint64_t n, const fg_tables* tables, const fg_gelf_extra* extra, uint8_t* d_out, uint64_t out_cap,
                          uint64_t* d_out_offsets, uint64_t* total, void* stream) {
    fg_encode_cfg ec{};
    ec.encoder = FG_ENC_GELF;
    ec.merger = FG_MERGE_NONE;
    if (extra) {
        ec.n_extra = extra->n;
        ec.extra_keys = extra->keys;
        ec.extra_values = extra->values;
    }
    return fg_encode_device(ctx, src_fmt, &ec, d_bytes, nbytes, d_offsets, n, tables, d_out, out_cap, d_out_offsets, nullptr, total, stream);
}

const char* fg_encode_error_string(uint8_t st) {
    switch (st) {
        case fg::ES_OK:
        case fg::ES_DECODE_FAILED: return "";
        case fg::ES_5424_DATE: return "Failed to parse date";                                         // rfc5424_encoder.rs:46
        case fg::ES_5424_FORMAT: return "Failed to parse date as Rfc3339 format";                     // rfc5424_encoder.rs:52
        case fg::ES_3164_TS: return "Failed to parse unix timestamp in RFC3164 encoder";              // rfc3164_encoder.rs:53
        case fg::ES_PASSTHROUGH_EMPTY: return "Cannot output empty raw message";                      // passthrough_encoder.rs:47
    }
    return nullptr;
}

const char* const kErr3164[] = {
    "",
    "Malformed RFC3164 event: Invalid priority",               // rfc3164_decoder.rs:128-130
    "Invalid priority",                                        // :136
    "Malformed RFC3164 event: Invalid timestamp or hostname",  // :121
    "Invalid time format",                                     // :158
    "Unable to parse RFC3164 date with year",                  // :176
    "Unable to parse the date in RFC3164 decoder",             // :211
    "<the reference panics here: index out of bounds (rfc3164_decoder.rs:67)>",
};

static int upload_tz(fg_ctx* ctx) {
    DeviceGuard g(ctx->device);
    if (ctx->d_tz) {
        (void)hipFree(ctx->d_tz);
        ctx->d_tz = nullptr;
    }
    ctx->r3164 = fg::r3164::Cfg{};
    ctx->r3164.current_year = ctx->r3164_year;
    if (ctx->tz_names.empty()) return FG_OK;
    fg::r3164::TzIndex idx;  // hash index, reject masks, year hints (fg_tz_index.hpp)
    if (!idx.build(ctx->tz_names, ctx->tz_first, ctx->tz_start, ctx->tz_off, ctx->r3164_year)) return FG_ERR_ARG;
    FG_HIP(ctx, hipMalloc((void**)&ctx->d_tz, idx.blob.size()));
    FG_HIP(ctx, hipMemcpy(ctx->d_tz, idx.blob.data(), idx.blob.size(), hipMemcpyHostToDevice));
    ctx->r3164.tz = idx.view(ctx->d_tz);
    return FG_OK;
}

int fg_set_rfc3164(fg_ctx* ctx, const fg_rfc3164_cfg* cfg) {
    if (!ctx || !cfg) return FG_ERR_ARG;
    ctx->r3164_auto_year = cfg->current_year == FG_YEAR_NOW;
    ctx->r3164_year = ctx->r3164_auto_year ? utc_year_now() : cfg->current_year;
    ctx->tz_names.clear();
    ctx->tz_first.clear();
    ctx->tz_start.clear();
    ctx->tz_off.clear();
    if (cfg->tz && cfg->tz->n_zones) {
        const fg_tz_table* t = cfg->tz;
        if (!t->names || !t->zone_first || !t->utc_start || !t->utc_offset) return FG_ERR_ARG;
        for (uint32_t z = 0; z < t->n_zones; ++z) {
            if (!t->names[z] || t->zone_first[z + 1] <= t->zone_first[z]) return FG_ERR_ARG;
            if (z && strcmp(t->names[z - 1], t->names[z]) >= 0) return FG_ERR_ARG;  // sorted bytewise, unique
            ctx->tz_names.emplace_back(t->names[z]);
        }
        ctx->tz_first.assign(t->zone_first, t->zone_first + t->n_zones + 1);
        const uint32_t ne = t->zone_first[t->n_zones];
        ctx->tz_start.assign(t->utc_start, t->utc_start + ne);
        ctx->tz_off.assign(t->utc_offset, t->utc_offset + ne);
    }
    ctx->r3164_set = true;
    return upload_tz(ctx);
}

const char* fg_error_string(fg_format fmt, uint8_t status) {
    if (status == FG_ST_BAD_UTF8) return "Invalid UTF-8 input";  // line_splitter.rs:23, nul_splitter.rs:36
    switch (fmt) {
        case FG_RFC5424:
            return status < sizeof(kErr5424) / sizeof(*kErr5424) ? kErr5424[status] : nullptr;
        case FG_LTSV:
            return status < sizeof(kErrLtsv) / sizeof(*kErrLtsv) ? kErrLtsv[status] : nullptr;
        case FG_GELF:
            return status < sizeof(kErrGelf) / sizeof(*kErrGelf) ? kErrGelf[status] : nullptr;
        case FG_RFC3164:
            return status < sizeof(kErr3164) / sizeof(*kErr3164) ? kErr3164[status] : nullptr;
    }
    return nullptr;
}

int fg_shard_plan(const uint64_t* offsets, uint64_t n, uint32_t g, uint64_t* line_starts) {
    if (!line_starts || g == 0 || (n && !offsets)) return FG_ERR_ARG;
    line_starts[0] = 0;
    line_starts[g] = n;
    if (n == 0) {
        for (uint32_t k = 1; k < g; ++k) line_starts[k] = 0;
        return FG_OK;
    }
    const uint64_t b0 = offsets[0], total = offsets[n] - b0;
    uint64_t lo = 0;
    for (uint32_t k = 1; k < g; ++k) {
        // first line whose start is >= k/g of the bytes (binary search; offsets is non-decreasing)
        uint64_t target = b0 + (uint64_t)((__uint128_t)total * k / g);
        uint64_t a = lo, b = n;
        while (a < b) {
            uint64_t m = a + (b - a) / 2;
            if (offsets[m] < target) a = m + 1;
            else b = m;
        }
        line_starts[k] = a;
        lo = a;
    }
    return FG_OK;
}

}  // extern "C"
