// fg_capi.cpp -- host side of the C ABI declared in include/fg_hip.h.
//
// Owns: device selection (gfx950 only, no CPU fallback), the per-ctx HIP stream, staging
// buffers for the host-buffer entry point, HIP-event timing of the decode kernels.
// It mirrors what XDecoder::new(&Config) + Box<dyn Decoder+Send>::clone do in the reference
// (src/flowgger/mod.rs:413-422, src/flowgger/decoder/mod.rs:23-36): a ctx is built once from the
// configuration and cloned per connection thread.
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
                                 const uint8_t* line_bad, const fg_launch_opts* lo);
extern "C" uint64_t fg_stash_bytes(uint32_t blocks);
extern "C" int fg_launch_rfc3164(const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t n, const fg::DevTables* t,
                                 const fg::r3164::Cfg* cfg, uint32_t tile_cap, hipStream_t stream, uint32_t strip,
                                 const uint8_t* line_bad);
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
                                      uint8_t* d_out, hipStream_t stream);
extern "C" uint64_t fg_frame_scratch_bytes(uint64_t nbytes);
extern "C" int fg_launch_frame(const uint8_t* d_bytes, uint64_t nbytes, uint32_t delim, uint8_t* scratch, uint64_t* d_offsets,
                               uint8_t* d_bad, uint64_t cap, uint64_t** d_total_out, hipStream_t stream);
extern "C" uint64_t fg_frame_block_bytes(void);
extern "C" int fg_launch_frame_slice(const uint8_t* d_bytes, uint64_t nbytes, uint32_t delim, uint8_t* scratch, uint64_t* d_offsets,
                                     uint8_t* d_bad, uint64_t cap, uint64_t blk0, uint64_t blk1, uint64_t** d_total_out,
                                     hipStream_t stream);
extern "C" int fg_launch_ltsv(const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t n, const fg::DevTables* t,
                              const fg::LtsvDevCfg* cfg, uint64_t avg_len, hipStream_t stream, uint64_t* stash,
                              uint32_t stash_blocks, uint32_t strip, const uint8_t* line_bad, const fg_launch_opts* lo);
extern "C" int fg_launch_gelf(const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t n, const fg::DevTables* t,
                              uint64_t avg_len, hipStream_t stream, uint64_t* stash, uint32_t stash_blocks, uint32_t strip,
                              const uint8_t* line_bad, const fg_launch_opts* lo);

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
    uint32_t epoch = 0;             // launch counter of this ctx
    uint8_t* d_frame = nullptr;  // fg_frame_device scratch (delimiter / UTF-8 masks, block counts)
    uint64_t d_frame_cap = 0;
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
    return d;
}

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

extern "C" {

int fg_abi_version(void) { return FG_ABI_VERSION; }

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
    if (ctx->d_frame) (void)hipFree(ctx->d_frame);
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

int fg_last_kernel_ms(fg_ctx* ctx, float* ms) {
    if (!ctx || !ms) return FG_ERR_ARG;
    *ms = 0.f;
    if (!ctx->timing || !ctx->ev_valid) return FG_OK;
    DeviceGuard g(ctx->device);
    FG_HIP(ctx, hipEventSynchronize(ctx->ev1));
    FG_HIP(ctx, hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
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
    uint64_t* d_total = nullptr;
    int lrc = fg_launch_frame(d_bytes, nbytes, framing == FG_FRAME_LINE ? 0x0Au : 0x00u, ctx->d_frame, d_offsets, d_bad_utf8,
                              cap_frames, &d_total, s);
    if (lrc != 0) {
        ctx->last_hip = lrc;
        return FG_ERR_HIP;
    }
    // frames = delimiters (+1 when the stream does not end with one)
    uint64_t total = 0;
    FG_HIP(ctx, hipMemcpyAsync(&total, d_total, 8, hipMemcpyDeviceToHost, s));
    FG_HIP(ctx, hipStreamSynchronize(s));
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

static int decode_frames_impl(fg_ctx* ctx, fg_format fmt, fg_framing framing, const uint8_t* d_bytes, uint64_t nbytes,
                              const uint64_t* d_offsets, uint64_t n, const uint8_t* d_bad_utf8, const fg_tables* tables,
                              void* stream, bool reset_counter, uint64_t span_bytes, uint32_t lane = 0);

int fg_decode_frames_device(fg_ctx* ctx, fg_format fmt, fg_framing framing, const uint8_t* d_bytes, uint64_t nbytes,
                            const uint64_t* d_offsets, uint64_t n, const uint8_t* d_bad_utf8, const fg_tables* tables,
                            void* stream) {
    return decode_frames_impl(ctx, fmt, framing, d_bytes, nbytes, d_offsets, n, d_bad_utf8, tables, stream, true, nbytes);
}

// reset_counter = false: a further slice of a batch whose entry counter is already live (the
// pipelined host path decodes one batch as several slices on two streams).  span_bytes = the bytes
// the n lines cover (the launch geometry is planned from the average line length; nbytes is only
// the readable range of d_bytes and, for a slice, covers the whole batch).
static int decode_frames_impl(fg_ctx* ctx, fg_format fmt, fg_framing framing, const uint8_t* d_bytes, uint64_t nbytes,
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
    ctx->epoch += 1u;
    if (ctx->epoch == 0u) ctx->epoch = 1u;  // (0 = the ring's initial content: never a valid epoch)
    dt.epoch = ctx->epoch;
    dt.pending = ctx->d_pending + (ctx->epoch % kPendingRing);
    if (ctx->timing) FG_HIP(ctx, hipEventRecord(ctx->ev0, s));
    int rc;
    const uint64_t avg_len = (span_bytes + n - 1) / n;
    switch (fmt) {
        case FG_RFC5424:
            rc = fg_launch_rfc5424(d_bytes, d_offsets, n, &dt, avg_len, s, stash, ctx->stash_blocks, (uint32_t)framing,
                                   d_bad_utf8, &ctx->lo);
            break;
        case FG_LTSV:
            rc = fg_launch_ltsv(d_bytes, d_offsets, n, &dt, &ctx->ltsv, avg_len, s, stash, ctx->stash_blocks,
                                (uint32_t)framing, d_bad_utf8, &ctx->lo);
            break;
        case FG_GELF:
            rc = fg_launch_gelf(d_bytes, d_offsets, n, &dt, avg_len, s, stash, ctx->stash_blocks,
                                (uint32_t)framing, d_bad_utf8, &ctx->lo);
            break;
        case FG_RFC3164:
            if (!ctx->r3164_set) return FG_ERR_ARG;  // fg_set_rfc3164 first
            if ((rc = refresh_year(ctx)) != FG_OK) return rc;
            rc = fg_launch_rfc3164(d_bytes, d_offsets, n, &dt, &ctx->r3164, pick_tile_cap(ctx, span_bytes, n, 56 * 1024), s, (uint32_t)framing,
                                   d_bad_utf8);
            break;
        default:
            return FG_ERR_UNSUPPORTED;
    }
    if (rc != 0) {
        ctx->last_hip = rc;
        return FG_ERR_HIP;
    }
    if (ctx->timing) {
        FG_HIP(ctx, hipEventRecord(ctx->ev1, s));
        ctx->ev_valid = true;
    }
    return FG_OK;
}

int fg_alloc_pinned(uint64_t bytes, void** out) {
    if (!out) return FG_ERR_ARG;
    *out = nullptr;
    return hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault) == hipSuccess ? FG_OK : FG_ERR_HIP;
}
void fg_free_pinned(void* p) {
    if (p) (void)hipHostFree(p);
}

// Streams and events of the pipelined host paths (created on first use): uploads (stream2), kernels (stream3) and downloads (the
// ctx's FIRST stream) each get their own stream, chained by events per slice.  Which stream carries which direction matters on
// this platform (tools/probe/stream_pairs.cpp, MI355X / ROCm 7.2): an H2D and a D2H copy run at the same time -- 97 GB/s for
// the two together -- only when one of the two streams is the first one created; any other pair shares a copy path and the two
// directions take turns (57 GB/s for both together).  fg_measure_link measures on exactly these streams.
// Chained by events per slice -- H2D copies run back to back on one stream (nothing else is ever queued between two of
// them), D2H copies on another, and the link carries both directions at once.  (Round 2 alternated whole slices -- H2D, kernel,
// D2H -- between two streams: measured, that gave the rate of NO overlap at all, 42 of 57 GB/s.)
static int ensure_pipeline(fg_ctx* ctx, uint32_t slices) {
    if (!ctx->stream2) {
        FG_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
        FG_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_ready, hipEventDisableTiming));
    }
    if (!ctx->stream3) FG_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream3, hipStreamNonBlocking));
    while (ctx->ev_slice.size() < 2ull * slices) {
        hipEvent_t e = nullptr;
        FG_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->ev_slice.push_back(e);
    }
    if (!ctx->s_up) {
        // The runtime binds its copy paths to streams as they FIRST copy, and which stream went first decides whether the uploads
        // and downloads below overlap (measured, tools/probe/stream_pairs.cpp and four orderings of this code: with the ctx's own
        // stream first, fg_decode_batch runs at 198 M lines/s; with stream2 / stream3 first, at 160 M).  So the ctx's own stream
        // copies a few bytes each way before the other two are ever used.
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
    {   // on the very streams the pipelined host paths copy on: uploads on stream2, downloads on the ctx's first stream
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

int fg_decode_batch(fg_ctx* ctx, fg_format fmt, const uint8_t* bytes, uint64_t nbytes, const uint64_t* offsets,
                    uint64_t n, fg_tables* out) {
    if (!ctx || !out || (n && !offsets) || (nbytes && !bytes)) return FG_ERR_ARG;
    if (n && (offsets[n] > nbytes || offsets[0] > offsets[n])) return FG_ERR_ARG;
    DeviceGuard g(ctx->device);
    int rc;
    const uint32_t slices = slice_count(nbytes, n);
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
        fg_span* hs[6] = {ht.hostname, ht.appname, ht.procid, ht.msgid, ht.msg, ht.full_msg};
        fg_span* ds[6] = {dt.hostname, dt.appname, dt.procid, dt.msgid, dt.msg, dt.full_msg};
        auto issue = [&](uint32_t k) -> int {
            const uint64_t l0 = cut[k], l1 = cut[k + 1], rows = l1 - l0;
            if (rows == 0) return FG_OK;
            // ---- upload: the slice's offsets (the first slice also takes offsets[0]) and its bytes, copied on 16-byte boundaries
            //      (the neighbouring bytes are the same data)
            const uint64_t o0 = k == 0 ? l0 : l0 + 1;
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
            const int drc = decode_frames_impl(ctx, fmt, FG_FRAME_NONE, ctx->d_bytes, nbytes, ctx->d_offsets + l0, rows, nullptr, &sl, (void*)s_run, false,
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

// fg_frame_decode_batch for a LARGE raw chunk: the chunk crosses the link in slices (multiples of the framing kernels' 16 KiB
// block) on the upload stream; as soon as a slice is there it is framed (delimiter ranks continue where the slice before stopped),
// its frame count comes back to the host through a pinned word, the frames that END in it are decoded, and their rows + offsets go
// back on the download stream -- all while the next slices are still on the link.  The host never frames, never uploads offsets.
// Tables are sized from what the ctx's last chunk held; a chunk that outgrows the estimate (or the entry table) returns
// FG_ERR_UNSUPPORTED and takes the one-piece path, which counts first.
static int frame_decode_sliced(fg_ctx* ctx, fg_format fmt, fg_framing framing, const uint8_t* bytes, uint64_t nbytes, int final,
                               fg_tables* out, const uint64_t** out_offsets, uint64_t* n_frames, uint64_t* consumed) {
    int rc;
    const uint64_t blk = fg_frame_block_bytes();
    uint64_t slice = nbytes / 8;
    if (slice < (8ull << 20)) slice = 8ull << 20;
    if (slice > (32ull << 20)) slice = 32ull << 20;
    slice = slice / blk * blk;
    const uint32_t slices = (uint32_t)((nbytes + slice - 1) / slice);
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
    if (slices + 2 > 4096) return FG_ERR_UNSUPPORTED;  // (h_cnt: frame counts in the first half, entry counters in the second)
    uint64_t tab_bytes = 0;
    carve(nullptr, cap, ent_cap, nullptr, &tab_bytes);
    if ((rc = grow_dev(ctx, (void**)&ctx->d_tab, &ctx->d_tab_cap, tab_bytes)) != FG_OK) return rc;
    if ((rc = grow_pinned(ctx, (void**)&ctx->h_tab, &ctx->h_tab_cap, tab_bytes)) != FG_OK) return rc;
    fg_tables dt, ht;
    carve(ctx->d_tab, cap, ent_cap, &dt, nullptr);
    carve(ctx->h_tab, cap, ent_cap, &ht, nullptr);
    FG_HIP(ctx, hipMemsetAsync(dt.ent_used, 0, 8, s_run));
    FG_HIP(ctx, hipMemsetAsync(ctx->d_bad, 0, cap + 1, s_run));
    const uint32_t delim = framing == FG_FRAME_LINE ? 0x0Au : 0x00u;
    std::vector<hipEvent_t>& ev = ctx->ev_slice;
    // every upload is queued NOW (they depend on nothing): the link never waits for the host
    for (uint32_t k = 0; k < slices; ++k) {
        const uint64_t b0 = (uint64_t)k * slice, b1 = k + 1 == slices ? nbytes : b0 + slice;
        FG_HIP(ctx, hipMemcpyAsync(ctx->d_bytes + b0, bytes + b0, b1 - b0, hipMemcpyHostToDevice, s_up));
        if (k + 1 == slices) FG_HIP(ctx, hipMemsetAsync(ctx->d_bytes + nbytes, 0, up(nbytes, 16) + 16 - nbytes, s_up));
        FG_HIP(ctx, hipEventRecord(ev[2 * k], s_up));
    }
    // frame slice k once it is there, bring its cumulative frame count back (all asynchronous)
    auto enqueue = [&](uint32_t k) -> int {
        const uint64_t b0 = (uint64_t)k * slice, b1 = k + 1 == slices ? nbytes : b0 + slice;
        FG_HIP(ctx, hipStreamWaitEvent(s_run, ev[2 * k], 0));
        uint64_t* d_total = nullptr;
        const uint64_t blk0 = b0 / blk, blk1 = k + 1 == slices ? nblk_total : b1 / blk;
        const int lrc = fg_launch_frame_slice(ctx->d_bytes, nbytes, delim, ctx->d_frame, ctx->d_offsets, ctx->d_bad, cap, blk0, blk1, &d_total, s_run);
        if (lrc != 0) {
            ctx->last_hip = lrc;
            return FG_ERR_HIP;
        }
        FG_HIP(ctx, hipMemcpyAsync(ctx->h_cnt + k, d_total, 8, hipMemcpyDeviceToHost, s_run));
        FG_HIP(ctx, hipEventRecord(ev[2 * k + 1], s_run));
        return FG_OK;
    };
    uint64_t done = 0;  // frames decoded so far
    // The entry columns come back per slice as well: after a slice's decode the entry counter goes into a pinned word, and two
    // iterations later -- the frame-count event the host waits for then was queued behind it -- the entries between two counter
    // values are copied on the download stream, while later slices are still on the link.  RFC5424 only: measured (profiles/
    // r03y_e2e_*.json, 4 M lines per call) the structured-data corpus goes from 63 to 90 M lines/s with it, while the GELF and LTSV
    // corpora got SLOWER on this path (78 -> 46, 108 -> 87 M lines/s; not understood yet) -- their entries come back at the end.
    const bool early = fmt == FG_RFC5424;
    uint64_t* const ent_cnt = ctx->h_cnt + 4096;
    uint64_t ent_done = 0;
    auto count_entries = [&](uint32_t k) -> int {
        if (early) FG_HIP(ctx, hipMemcpyAsync(ent_cnt + k, dt.ent_used, 8, hipMemcpyDeviceToHost, s_run));
        return FG_OK;
    };
    auto download_entries = [&](uint64_t e1) -> int {  // entries [ent_done, e1)
        if (e1 <= ent_done) return FG_OK;
        const uint64_t e0 = ent_done, m = e1 - e0;
        FG_HIP(ctx, hipMemcpyAsync(ht.ent_name + e0, dt.ent_name + e0, m * 8, hipMemcpyDeviceToHost, s_down));
        FG_HIP(ctx, hipMemcpyAsync(ht.ent_val + e0, dt.ent_val + e0, m * 8, hipMemcpyDeviceToHost, s_down));
        FG_HIP(ctx, hipMemcpyAsync(ht.ent_type + e0, dt.ent_type + e0, m, hipMemcpyDeviceToHost, s_down));
        FG_HIP(ctx, hipMemcpyAsync(ht.ent_flags + e0, dt.ent_flags + e0, m, hipMemcpyDeviceToHost, s_down));
        ent_done = e1;
        return FG_OK;
    };
    auto decode_rows = [&](uint64_t f0, uint64_t f1, uint64_t span_bytes) -> int {  // frames [f0, f1): decode + download
        if (f1 == f0) return FG_OK;
        const uint64_t rows = f1 - f0;
        fg_tables sl = dt;
        sl.n = rows;
        sl.meta += f0; sl.ts += f0; sl.hostname += f0; sl.appname += f0; sl.procid += f0; sl.msgid += f0; sl.msg += f0; sl.full_msg += f0;
        sl.ent_first += f0; sl.ent_count += f0;
        int r = decode_frames_impl(ctx, fmt, framing, ctx->d_bytes, nbytes, ctx->d_offsets + f0, rows, ctx->d_bad + f0, &sl, (void*)s_run, false, span_bytes);
        if (r != FG_OK) return r;
        FG_HIP(ctx, hipEventRecord(ctx->ev_ready, s_run));
        FG_HIP(ctx, hipStreamWaitEvent(s_down, ctx->ev_ready, 0));
        FG_HIP(ctx, hipMemcpyAsync(ctx->h_off + f0 + (f0 ? 1 : 0), ctx->d_offsets + f0 + (f0 ? 1 : 0), (rows + (f0 ? 0 : 1)) * 8, hipMemcpyDeviceToHost, s_down));
        FG_HIP(ctx, hipMemcpyAsync(ht.meta + f0, dt.meta + f0, rows * 4, hipMemcpyDeviceToHost, s_down));
        FG_HIP(ctx, hipMemcpyAsync(ht.ts + f0, dt.ts + f0, rows * 8, hipMemcpyDeviceToHost, s_down));
        fg_span* hs[6] = {ht.hostname, ht.appname, ht.procid, ht.msgid, ht.msg, ht.full_msg};
        fg_span* ds[6] = {dt.hostname, dt.appname, dt.procid, dt.msgid, dt.msg, dt.full_msg};
        for (int j = 0; j < 6; ++j) FG_HIP(ctx, hipMemcpyAsync(hs[j] + f0, ds[j] + f0, rows * 8, hipMemcpyDeviceToHost, s_down));
        FG_HIP(ctx, hipMemcpyAsync(ht.ent_first + f0, dt.ent_first + f0, rows * 4, hipMemcpyDeviceToHost, s_down));
        FG_HIP(ctx, hipMemcpyAsync(ht.ent_count + f0, dt.ent_count + f0, rows * 4, hipMemcpyDeviceToHost, s_down));
        return FG_OK;
    };
    if ((rc = enqueue(0)) != FG_OK) {
        drain();
        return rc;
    }
    for (uint32_t k = 0; k < slices; ++k) {
        if (k + 1 < slices && (rc = enqueue(k + 1)) != FG_OK) {
            drain();
            return rc;
        }
        FG_HIP(ctx, hipEventSynchronize(ev[2 * k + 1]));
        const uint64_t total = ctx->h_cnt[k];  // delimiters up to the end of slice k = frames that are complete
        if (early && k >= 2) {  // the decode of slice k - 2 and its counter copy were queued before this slice's framing: both are done
            const uint64_t cnt = ent_cnt[k - 2];
            if (cnt > ent_cap) {
                drain();
                return FG_ERR_UNSUPPORTED;
            }
            if ((rc = download_entries(cnt)) != FG_OK) {
                drain();
                return rc;
            }
        }
        if (total + 1 > cap) {
            drain();
            ctx->frames_per_byte = (double)(total + 1) / (double)(((uint64_t)k + 1) * slice);
            return FG_ERR_UNSUPPORTED;
        }
        const uint64_t b1 = k + 1 == slices ? nbytes : ((uint64_t)k + 1) * slice;
        if (k + 1 < slices) {
            if ((rc = decode_rows(done, total, b1 - (uint64_t)k * slice)) != FG_OK || (rc = count_entries(k)) != FG_OK) {
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
    uint64_t used = 0;
    FG_HIP(ctx, hipMemcpyAsync(&used, dt.ent_used, 8, hipMemcpyDeviceToHost, s_run));
    FG_HIP(ctx, hipStreamSynchronize(s_run));
    if (done) ctx->frames_per_byte = (double)done / (double)nbytes;
    if (used > ent_cap) {
        drain();
        return FG_ERR_UNSUPPORTED;
    }
    if ((rc = download_entries(used)) != FG_OK) {  // what the last two slices appended
        drain();
        return rc;
    }
    *ht.ent_used = used;
    drain();
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

int fg_frame_decode_batch(fg_ctx* ctx, fg_format fmt, fg_framing framing, const uint8_t* bytes, uint64_t nbytes, int final,
                          fg_tables* out, const uint64_t** out_offsets, uint64_t* n_frames, uint64_t* consumed) {
    if (!ctx || !out || !out_offsets || !n_frames || !consumed || (nbytes && !bytes)) return FG_ERR_ARG;
    if (framing != FG_FRAME_LINE && framing != FG_FRAME_NUL) return FG_ERR_UNSUPPORTED;
    if (nbytes >= (48ull << 20) && !(ctx->lo.flags & FG_LO_TRANSCODE_ONE_PIECE)) {
        DeviceGuard g(ctx->device);
        *n_frames = 0;
        *consumed = 0;
        *out_offsets = nullptr;
        const int rc = frame_decode_sliced(ctx, fmt, framing, bytes, nbytes, final, out, out_offsets, n_frames, consumed);
        if (rc != FG_ERR_UNSUPPORTED) return rc;
    }
    return frame_decode_one_piece(ctx, fmt, framing, bytes, nbytes, final, out, out_offsets, n_frames, consumed);
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
        uint64_t* d_total = nullptr;
        int lrc = fg_launch_frame(ctx->d_bytes, nbytes, framing == FG_FRAME_LINE ? 0x0Au : 0x00u, ctx->d_frame, ctx->d_offsets,
                                  ctx->d_bad, cap, &d_total, s);
        if (lrc != 0) {
            ctx->last_hip = lrc;
            return FG_ERR_HIP;
        }
        FG_HIP(ctx, hipMemcpyAsync(&total, d_total, 8, hipMemcpyDeviceToHost, s));
        FG_HIP(ctx, hipStreamSynchronize(s));
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
    if (!ctx->stream2) {
        FG_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
        FG_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_ready, hipEventDisableTiming));
    }
    hipStream_t lanes[2] = {ctx->stream, ctx->stream2};
    uint32_t slices = (uint32_t)(nbytes / (32ull << 20));
    if (slices < 2) slices = 2;
    if (slices > 64) slices = 64;
    if (n < slices) return FG_ERR_UNSUPPORTED;
    std::vector<uint64_t> cut(slices + 1);
    if (fg_shard_plan(offsets, n, slices, cut.data()) != FG_OK) return FG_ERR_ARG;
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
        (void)hipStreamSynchronize(lanes[0]);
        (void)hipStreamSynchronize(lanes[1]);
    };
    // queue a slice's upload, decode and count kernel
    auto enqueue = [&](uint32_t k) -> int {
        hipStream_t s = lanes[k & 1u];
        const uint64_t l0 = cut[k], l1 = cut[k + 1];
        if (l1 == l0) return FG_OK;
        const uint64_t b0 = offsets[l0] & ~15ull, b1 = offsets[l1];
        if (b1 > b0) FG_HIP(ctx, hipMemcpyAsync(ctx->d_bytes + b0, bytes + b0, b1 - b0, hipMemcpyHostToDevice, s));
        const fg_tables sl = sl_tables(l0, l1);
        // (the two lanes' kernels may be in flight at the same time: each lane parks its entries in its own stash)
        int r = decode_frames_impl(ctx, fmt, FG_FRAME_NONE, ctx->d_bytes, nbytes, ctx->d_offsets + l0, l1 - l0, nullptr, &sl, (void*)s, false,
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
        return FG_OK;
    };
    for (uint32_t k = 0; k < slices; ++k) {
        if (k == 0 && (rc = enqueue(0)) != FG_OK) {
            drain();
            return rc;
        }
        if (k + 1 < slices && (rc = enqueue(k + 1)) != FG_OK) {
            drain();
            return rc;
        }
        // ---- finish slice k: offsets from `base`, its size, the write kernel, the download ----
        hipStream_t s = lanes[k & 1u];
        const uint64_t l0 = cut[k], l1 = cut[k + 1], rows = l1 - l0;
        if (rows == 0) continue;
        if (fg_launch_encode_scan(d_sizes + l0, d_block_sums + blocks_before(k), rows, d_out_offsets + l0, base, s) != 0) {
            drain();
            return FG_ERR_HIP;
        }
        uint64_t end = 0;
        FG_HIP(ctx, hipMemcpyAsync(&end, d_out_offsets + l1, 8, hipMemcpyDeviceToHost, s));
        FG_HIP(ctx, hipStreamSynchronize(s));
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
        if (fg_launch_encode_write(ctx->d_bytes, ctx->d_offsets + l0, rows, &sdt, &cfg, tile_cap, cfg_lds, d_out_offsets + l0, ctx->d_tout, s) != 0) {
            drain();
            return FG_ERR_HIP;
        }
        uint8_t* hh = ctx->h_tout;
        if (end > base) FG_HIP(ctx, hipMemcpyAsync(hh + o_msgs + base, ctx->d_tout + base, end - base, hipMemcpyDeviceToHost, s));
        FG_HIP(ctx, hipMemcpyAsync(hh + o_offs + l0 * 8, d_out_offsets + l0, (rows + 1) * 8, hipMemcpyDeviceToHost, s));
        FG_HIP(ctx, hipMemcpyAsync(hh + o_meta + l0 * 4, dt.meta + l0, rows * 4, hipMemcpyDeviceToHost, s));
        FG_HIP(ctx, hipMemcpyAsync(hh + o_st + l0, d_enc_status + l0, rows, hipMemcpyDeviceToHost, s));
        base = end;
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
    if (framing == FG_FRAME_NONE && nbytes >= (64ull << 20) && n >= 4096 && !(ctx->lo.flags & FG_LO_TRANSCODE_ONE_PIECE)) {
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
    lrc = fg_launch_encode_write(d_bytes, d_offsets, n, &dt, &cfg, tile_cap, cfg_lds, d_out_offsets, d_out, s);
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
