/*
 * fg_hip.h -- C ABI of libfg_hip: the MI355X (gfx950) bulk log-line decoder that replaces
 * flowgger's per-line Decoder::decode() hot path.
 *
 * Reference interface replaced (paths relative to the flowgger source tree):
 *   trait Decoder { fn decode(&self, line:&str) -> Result<Record,&'static str> }
 *                                                   src/flowgger/decoder/mod.rs:44-46
 *   RFC5424Decoder::decode                          src/flowgger/decoder/rfc5424_decoder.rs:17-50
 *   LTSVDecoder::new / decode                       src/flowgger/decoder/ltsv_decoder.rs:23-221
 *   GelfDecoder::decode                             src/flowgger/decoder/gelf_decoder.rs:34-125
 *   RFC3164Decoder::decode                          src/flowgger/decoder/rfc3164_decoder.rs:31-213
 *   Record / StructuredData / SDValue               src/flowgger/record.rs:3-82
 *   the per-line call site the batching framer replaces
 *                                                   src/flowgger/splitter/line_splitter.rs:44-54
 *
 * Model: the framer packs N framed lines (valid UTF-8, framing bytes already stripped exactly
 * as BufRead::lines()/split(0)/syslen do) into one byte buffer + an offset array and makes ONE
 * call; hand-written HIP kernels tokenise every line and emit field-offset TABLES (struct of
 * arrays below).  A `Record` identical to the reference decoder's is materialised from a table
 * row + the line bytes (fg_tables_serialize / the Rust shim in INTEGRATION.md).  Errors are
 * part of the result: status[i] != 0 indexes the reference's exact &'static str.
 *
 * Plain C, plain pointers and sizes; no torch / C++ types.  All functions return 0 on success
 * or a negative FG_ERR_* (never throw, never abort).
 */
#ifndef FG_HIP_H
#define FG_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FG_ABI_VERSION 4 /* 4 (round 6): fg_frame_decode_device (framing inside the decode kernels), FG_LO_NO_FUSED_FRAMING, fg_launch_opts.fused_look / fused_ext (the struct grew), fg_last_host_path; 3 (round 5): fg_launch_opts.ent_chunk (the struct grew), FG_LO_STATIC_CHUNKS / _FRAME_SELFTEST_STALL, fg_ticket_ring_check; 2 (round 4): FG_YEAR_NOW = INT32_MIN, FG_F_LTSV_NOVALUE and the failed-LTSV-row count, ent_used = RESERVED slots, fg_calibrate_device */

typedef enum fg_format { FG_RFC5424 = 0, FG_LTSV = 1, FG_GELF = 2, FG_RFC3164 = 3 } fg_format;

/* return codes */
enum {
    FG_OK = 0,
    FG_ERR_ARG = -1,       /* NULL / inconsistent arguments */
    FG_ERR_HIP = -2,       /* a HIP runtime call failed (fg_last_hip_error has the code) */
    FG_ERR_NO_DEVICE = -3, /* no gfx950 device / kernels not loadable: there is NO CPU fallback */
    FG_ERR_ENT_OVERFLOW = -4, /* entry table too small; tables are valid except status==FG_ST_OVERFLOW rows */
    FG_ERR_UNSUPPORTED = -5,
    FG_ERR_NOMEM = -6      /* host allocation failed */
};

/* SDValue discriminants (record.rs:3-11) + the SD-element marker used in the entry table */
enum {
    FG_T_STRING = 0, FG_T_BOOL = 1, FG_T_F64 = 2, FG_T_I64 = 3, FG_T_U64 = 4, FG_T_NULL = 5,
    FG_T_SDID = 6 /* entry opens a new StructuredData element; name = its sd_id */
};

/* A byte span inside ONE line: off is relative to the line's first byte (offsets[i]).
 * len == FG_NONE means Option::None. */
#define FG_NONE 0xFFFFFFFFu
typedef struct fg_span { uint32_t off; uint32_t len; } fg_span;

/* meta word: status | facility<<8 | severity<<16 | flags<<24
 *   status   0 = Ok, else the format's error index (fg_error_string)
 *   facility / severity  0xFF = None
 *   flags    FG_F_* */
#define FG_META_STATUS(m)   ((uint8_t)((m) & 0xFF))
#define FG_META_FACILITY(m) ((uint8_t)(((m) >> 8) & 0xFF))
#define FG_META_SEVERITY(m) ((uint8_t)(((m) >> 16) & 0xFF))
#define FG_META_FLAGS(m)    ((uint8_t)(((m) >> 24) & 0xFF))
enum {
    FG_F_TS_NOW = 1,        /* GELF without "timestamp": Record.ts = wall clock at materialisation (gelf_decoder.rs:109) */
    FG_F_HOST_ESC = 2,      /* GELF: hostname span holds JSON escapes (decode when materialising) */
    FG_F_MSG_ESC = 4,       /* GELF: short_message span holds JSON escapes */
    FG_F_FULLMSG_ESC = 8,   /* GELF: full_message span holds JSON escapes */
    FG_F_BOM = 16,          /* RFC5424: line started with U+FEFF (spans already skip it) */
    FG_F_GELF_RETRY = 32,   /* GELF: accepted via the '\n' -> "\\n" retry (gelf_decoder.rs:44-46); when decoding
                               escapes a backslash followed by a raw LF means backslash + 'n' */
    FG_F_MSG_JOIN = 64,     /* RFC3164 standard form: Record.msg = the msg span's whitespace-separated tokens joined
                               with single spaces (`_log_tokens[1..].join(" ")`, rfc3164_decoder.rs:70) */
    FG_F_LTSV_NOVALUE = 128 /* LTSV: the line has tab-separated parts without ':'; the reference println!s
                               "Missing value for name '{}'" for each while it decodes (ltsv_decoder.rs:99) -- set on Ok AND on
                               failed rows (a failed row's hostname.off = how many parts had been reported when the decode
                               stopped; the same number, saturated at 254, is in its facility byte for callers that only
                               see the meta column).  fg_tables_stdout reproduces the text. */
};
#define FG_ST_OVERFLOW 0xFE /* status: the line's entries did not get slots: the table (ent_cap) is used up, counting the slots
                               parked in other waves' reservations (see ent_used).  Re-run with more; all other rows are valid */
#define FG_ST_BAD_UTF8 0xFD /* status: the frame is not valid UTF-8; the reference never decodes it, it prints
                               "Invalid UTF-8 input" and drops it (line_splitter.rs:22-25, nul_splitter.rs:34-40) */

/* How the frames handed to the decoders are delimited inside the byte buffer. */
typedef enum fg_framing {
    FG_FRAME_NONE = 0, /* offsets delimit bare lines (framing bytes already stripped by the caller) */
    FG_FRAME_LINE = 1, /* BufRead::lines(): "\n" terminated, the "\n" and one preceding "\r" are not part of the line */
    FG_FRAME_NUL = 2   /* BufRead::split(0): "\0" terminated */
} fg_framing;

/* entry flags */
enum {
    FG_EF_VAL_ESC = 1,   /* value span needs unescaping: RFC5424 \" \\ \] (rfc5424_decoder.rs:105-125) or JSON escapes (GELF) */
    FG_EF_NAME_ESC = 2,  /* GELF: key span holds JSON escapes */
    FG_EF_SUFFIX = 4     /* LTSV: append the configured type suffix to the name (ltsv_decoder.rs:131-136) */
};

/* One entry = one (name, SDValue) pair or one SD-element header, 18 bytes as SoA:
 *   ent_name[k]  span of the name WITHOUT the leading '_' the decoders add (sd_id for FG_T_SDID)
 *   ent_val[k]   String: span packed as off | (uint64)len<<32 ; Bool: 0/1 ; F64: IEEE bits ;
 *                I64: two's complement ; U64: value ; Null/SDID: 0
 *   ent_type[k]  FG_T_*      ent_flags[k]  FG_EF_*
 * Line i owns entries [ent_first[i], ent_first[i]+ent_count[i]) in decoder order.  Slices of
 * different lines may appear in any order inside the entry table (wave-level allocation). */
typedef struct fg_tables {
    uint64_t n;          /* rows (lines) */
    uint64_t ent_cap;    /* capacity of the ent_* arrays, in entries */
    uint32_t* meta;      /* [n] */
    double*   ts;        /* [n] Record.ts, bit-exact (utils/mod.rs:23-28) */
    fg_span*  hostname;  /* [n] */
    fg_span*  appname;   /* [n] */
    fg_span*  procid;    /* [n] */
    fg_span*  msgid;     /* [n] */
    fg_span*  msg;       /* [n] */
    fg_span*  full_msg;  /* [n] */
    uint32_t* ent_first; /* [n] */
    uint32_t* ent_count; /* [n] */
    fg_span*  ent_name;  /* [ent_cap] */
    uint64_t* ent_val;   /* [ent_cap] */
    uint8_t*  ent_type;  /* [ent_cap] */
    uint8_t*  ent_flags; /* [ent_cap] */
    uint64_t* ent_used;  /* [1] entry slots RESERVED by the call (may exceed ent_cap on overflow).  Waves reserve slots in
                            chunks (one atomic per chunk instead of one per line group), so this is >= the sum of ent_count and
                            the range [0, ent_used) contains slots no row refers to (uninitialised holes): always walk the
                            entries through ent_first / ent_count.  Sizing ent_cap: a table with fewer than 256 slots per
                            resident wave x 16 (ent_cap < ~8 M on an MI355X) gets exact reservations -- no slack needed
                            beyond the entries themselves; above that, up to ent_cap / 16 slots plus one line's entries per
                            chunk (at most 4096 slots) can be stranded: size from the input bytes (the host-buffer entry points use
                            nbytes / 16 for RFC5424, nbytes / 8 otherwise) and retry with ent_used + ent_used / 8 on
                            FG_ERR_ENT_OVERFLOW, as they do. */
} fg_tables;

/* Fixed table bytes written per line (meta 4 + ts 8 + 6 spans 48 + ent_first/count 8). */
#define FG_ROW_BYTES 68u
#define FG_ENT_BYTES 18u

/* LTSVDecoder::new configuration (input.ltsv_schema / input.ltsv_suffixes, ltsv_decoder.rs:24-84).
 * Names are matched byte-exactly; types are FG_T_STRING..FG_T_U64. */
typedef struct fg_cfg {
    uint32_t n_schema;
    const char* const* schema_names;
    const uint8_t* schema_types;
    const char* suffix_bool; /* NULL = None */
    const char* suffix_f64;
    const char* suffix_i64;
    const char* suffix_u64;
} fg_cfg;

/* RFC3164Decoder configuration.  The reference takes two things from its environment that are configuration here:
 *   current_year  OffsetDateTime::now_utc().year() prepended to dates without a year (rfc3164_decoder.rs:179): a fixed
 *                 year (tests, replays), or FG_YEAR_NOW = what the reference does -- the library re-reads the UTC year
 *                 at every FG_RFC3164 decode call, so a long-lived decoder (and its clones) crosses New Year correctly
 *   tz            the IANA zone table behind time_tz::timezones::get_by_name (rfc3164_decoder.rs:195; the time-tz
 *                 crate embeds the tz database at build time): n_zones names SORTED BYTEWISE (exact match), zone i
 *                 owns entries [zone_first[i], zone_first[i+1]) of (utc_start, utc_offset): utc_offset[k] seconds east
 *                 of UTC are in effect from utc_start[k] on, the zone's first entry starts at INT64_MIN.  NULL = no
 *                 zone name is recognised.  flowgger_amd/tzdb.py builds it from the system's TZif files. */
typedef struct fg_tz_table {
    uint32_t n_zones;
    const char* const* names;
    const uint32_t* zone_first;
    const int64_t* utc_start;
    const int32_t* utc_offset;
} fg_tz_table;
/* current_year: follow the wall clock (UTC year re-read at every FG_RFC3164 decode call).  Outside every representable year
 * (INT32_MIN), so that year 0 -- which `time` accepts -- can be configured explicitly. */
#define FG_YEAR_NOW (-2147483647 - 1)
typedef struct fg_rfc3164_cfg {
    int32_t current_year;
    const fg_tz_table* tz;
} fg_rfc3164_cfg;

typedef struct fg_ctx fg_ctx;
#define FG_STREAM_OWN ((void*)(intptr_t)-1)

int fg_abi_version(void);

/* Launch-geometry overrides of a ctx: the parity sweeps over the kernel variants (tests/) and tuning (tools/) set them through
 * this call.  The library reads NO environment variables (a stray FG_* in a daemon's environment must not change how it runs);
 * a zero field = the library's own choice; results are identical for every setting.  Clones made afterwards inherit them.
 * opts == NULL restores the defaults. */
typedef struct fg_launch_opts {
    uint32_t lines_per_group; /* lines a wave takes per group, 1..64 */
    uint32_t tile_cap;        /* LDS tile bytes (rounded up to 1 KiB) */
    uint32_t waves_per_cu;    /* upper bound of resident waves per CU */
    uint32_t gelf_lds_budget; /* GELF: LDS bytes per wave the lines per group are fitted to */
    uint32_t gelf_window_kib; /* GELF: register prefetch window in KiB (2..6) */
    uint32_t flags;           /* FG_LO_* */
    uint32_t chunk_lines;     /* lines a wave takes at a time (the first chunk by block index, further ones drawn by ticket), 1..65536 */
    uint32_t ent_chunk;       /* entry slots a wave reserves from the table's counter at a time: 1 = exactly what each request needs,
                                 >= 2 = that many (tuning: fewer atomics on the one counter word against slots stranded at the end of
                                 every wave's last reservation) */
    uint32_t fused_look;      /* fused frame + decode launches (fg_frame_decode_device): bytes staged behind a tile, where its last line ends
                                 (a multiple of 16; the library's own choice: one average line) */
    uint32_t fused_ext;       /* ... bytes staged ON at a time when that line runs past them (a multiple of 16, at most 1024) */
} fg_launch_opts;
enum {
    FG_LO_GELF_GENERIC = 1,        /* GELF: the run-time-geometry kernel even where the constant-geometry instantiation applies */
    FG_LO_TRANSCODE_ONE_PIECE = 2, /* fg_transcode_batch / fg_frame_decode_batch: never slice a large batch over the streams */
    FG_LO_NO_HEAD = 4,             /* RFC5424: stage whole lines even when they are long (the head-only kernel is the default from 768 B) */
    FG_LO_FORCE_HEAD = 8,          /* RFC5424: the head-only kernel for lines of any length (parity sweeps) */
    FG_LO_SD_WALK = 16,            /* RFC5424: the lane-per-line structured-data walker even for long lines (the pair-parallel walk is the
                                      default from an average of 320 bytes per line) */
    FG_LO_SD_PAIRS = 32,           /* RFC5424: the pair-parallel structured-data walk for lines of any length (parity sweeps) */
    FG_LO_NO_ZERO_COPY = 64,       /* fg_decode_batch: the sliced hipMemcpy pipeline even when the caller's buffers are pinned (A/B, tests) */
    FG_LO_FRAME_KERNEL_UPLOAD = 128, /* fg_frame_decode_batch, pinned chunk: the framing scan reads the chunk in place and stores it to HBM
                                      itself instead of hipMemcpy uploads (measured slower on MI355X / ROCm 7.2: off by default) */
    FG_LO_FRAME_CLASSIC = 256,     /* framing: the three-kernel form (masks to HBM, one-workgroup scan, emit) instead of the one-pass chained
                                      scan -- which falls back to it by itself should its look-back ever give up (A/B, tests) */
    FG_LO_STATIC_CHUNKS = 512,     /* decode kernels: deal the chunks of a batch out round-robin over the waves (rounds 3-4) instead of by
                                      ticket -- every wave draws its next chunk from a per-launch counter when it needs one (A/B, tests) */
    FG_LO_FRAME_SELFTEST_STALL = 1024, /* framing self-test: the one-pass scan's second tile never publishes its descriptor, so the tiles behind
                                      it must give up within the spin bound and the call must come back through the classic kernels with
                                      the same result (tests; never set in production: it costs the spin bound, ~0.2 s) */
    FG_LO_NO_TAPER = 2048,         /* ticket launches: every chunk of the batch the same size -- by default the last chunk per wave's worth of
                                      lines is dealt out in halves, quarters and eighths of a chunk, so that the grid finishes together
                                      (fg_plan_policy.hpp; A/B, tests) */
    FG_LO_TAPER_1 = 4096,          /* ... at most one such level (halves) instead of the format's own depth; */
    FG_LO_TAPER_2 = 8192,          /* ... at most two (halves, quarters); both bits: three (tuning) */
    FG_LO_NO_FUSED_FRAMING = 16384, /* fg_frame_decode_batch / fg_transcode_batch: the separate framing pass (rounds 1-5) even where the decode kernels
                                      can frame a pinned chunk themselves (A/B, tests) */
    FG_LO_RFC3164_REGROUP = 32768, /* FG_RFC3164: hand the lines of the slow shapes (zone names, the custom form) to a second kernel, 64 of a kind to the workgroup,
                                      for batches of any size -- the library does so from 1 M lines on (tests) */
    FG_LO_RFC3164_NO_REGROUP = 65536, /* ... never: lines in arrival order, a wave runs the union of its lines' shapes (rounds 1-5; A/B) */
    FG_LO_RESERVED = 0x40000000    /* the library's own (fg_set_launch_opts clears it) */
};

/* Create a decoder context on HIP device `device` (replaces XDecoder::new(&Config),
 * flowgger/mod.rs:413-422).  cfg may be NULL (RFC5424 / GELF take no configuration).
 * Fails with FG_ERR_NO_DEVICE when no gfx950 GPU is usable -- there is no CPU fallback.
 * A ctx is cheap to clone per connection thread (decoder/mod.rs:23-36): fg_clone shares the
 * device-side configuration and gets its own stream + staging buffers. */
int fg_create(int device, const fg_cfg* cfg, fg_ctx** out);
int fg_clone(const fg_ctx* ctx, fg_ctx** out);
void fg_destroy(fg_ctx* ctx);
int fg_last_hip_error(const fg_ctx* ctx);
int fg_set_launch_opts(fg_ctx* ctx, const fg_launch_opts* opts);
/* Configure the RFC3164 decoder of this ctx (copied; clones made afterwards inherit it).  Required before the first
 * FG_RFC3164 decode (FG_ERR_ARG otherwise). */
int fg_set_rfc3164(fg_ctx* ctx, const fg_rfc3164_cfg* cfg);

/* Bytes of device scratch fg_decode_batch_device needs in `tables` are all caller-provided;
 * this helper returns the byte size of every array of an fg_tables for (n, ent_cap) so that a
 * caller can carve one allocation: sizes[k] for k = meta, ts, hostname, appname, procid, msgid,
 * msg, full_msg, ent_first, ent_count, ent_name, ent_val, ent_type, ent_flags, ent_used. */
#define FG_TABLE_ARRAYS 15
int fg_tables_layout(uint64_t n, uint64_t ent_cap, uint64_t sizes[FG_TABLE_ARRAYS]);

/* DEVICE-RESIDENT decode (the hot path): every pointer is a device pointer on ctx's device.
 *   d_bytes    packed lines; must be 16-byte aligned and readable up to nbytes rounded up to 16
 *   d_offsets  n+1 entries, offsets[0] >= 0, offsets[n] <= nbytes, non-decreasing
 *   tables     struct (host memory) of DEVICE array pointers, n rows, ent_cap entries
 *   stream     a hipStream_t passed through verbatim (NULL = HIP's default/null stream, which is
 *              what PyTorch's default current stream is), or FG_STREAM_OWN = the ctx's own
 *              non-blocking stream; the call is asynchronous on it.
 * replaces: `for line { decoder.decode(line) }` (line_splitter.rs:17,50). */
int fg_decode_batch_device(fg_ctx* ctx, fg_format fmt, const uint8_t* d_bytes, uint64_t nbytes,
                           const uint64_t* d_offsets, uint64_t n, const fg_tables* tables,
                           void* stream);

/* GPU FRAMING + UTF-8 VALIDATION of a raw byte stream (replaces `buf_reader.lines()` /
 * `buf_reader.split(0)` + `str::from_utf8`, splitter/line_splitter.rs:17-25, nul_splitter.rs:18-40):
 *   d_bytes     raw stream chunk (device), 16-byte aligned, readable up to nbytes rounded up to 16
 *   d_offsets   out, cap_frames + 2 entries: frame i = [offsets[i], offsets[i+1]) INCLUDING its
 *               terminator; a final unterminated piece is a frame if it is non-empty
 *   d_bad_utf8  out, cap_frames + 1 bytes: 1 = the frame is not valid UTF-8
 *   n_frames    out (host): number of frames; the call synchronises the stream to read it
 * Returns FG_ERR_ENT_OVERFLOW when cap_frames is too small (*n_frames then holds the need).
 * framing must be FG_FRAME_LINE or FG_FRAME_NUL.  (syslen framing is a sequential prefix chain
 * per connection and stays on the host: flowgger_amd/host/fg_decoder.hpp.) */
int fg_frame_device(fg_ctx* ctx, fg_framing framing, const uint8_t* d_bytes, uint64_t nbytes,
                    uint64_t* d_offsets, uint8_t* d_bad_utf8, uint64_t cap_frames, uint64_t* n_frames,
                    void* stream);

/* Decode frames as produced by fg_frame_device: the kernels strip the terminators themselves, and
 * frames flagged in d_bad_utf8 (may be NULL) get status FG_ST_BAD_UTF8 instead of a decode.
 * fg_decode_batch_device(...) == fg_decode_frames_device(..., FG_FRAME_NONE, ..., NULL, ...). */
int fg_decode_frames_device(fg_ctx* ctx, fg_format fmt, fg_framing framing, const uint8_t* d_bytes,
                            uint64_t nbytes, const uint64_t* d_offsets, uint64_t n,
                            const uint8_t* d_bad_utf8, const fg_tables* tables, void* stream);

/* FRAME + DECODE IN ONE KERNEL (round 6; ABI 4): the decode kernel frames the raw stream chunk itself -- one read of the stream, no
 * framing pass, no frame count that visits the host between two kernels.  Replaces, for a chunk of the stream, the whole loop of
 * LineSplitter::run / NulSplitter::run (splitter/line_splitter.rs:17-54, nul_splitter.rs:18-60): `lines()` / `split(0)`, the UTF-8
 * check and decoder.decode(line) for every frame.
 *   d_bytes      raw stream chunk (device-addressable: HBM, or the device view of pinned host memory), 16-byte aligned, readable
 *                up to nbytes rounded up to 16
 *   final        nonzero: the stream ends with this chunk (an unterminated last piece is a frame); zero: it is not a frame, and
 *                d_offsets[frames] says where it starts (carry it over)
 *   d_offsets    out, cap_frames + 2 entries: frame i = [d_offsets[i], d_offsets[i + 1]) INCLUDING its terminator
 *   tables       device arrays for cap_frames rows (tables->n >= cap_frames); a frame that is not valid UTF-8 gets FG_ST_BAD_UTF8
 *   avg_line_hint  the average frame length to plan the launch for (0 = what this ctx last saw); only the speed depends on it
 *   d_result     out (device), two words: [0] the frames of the chunk -- when it exceeds cap_frames the rows beyond were not
 *                written: run again with more; [1] nonzero: the kernel's look-back gave up (never seen; bounded spin) -- nothing is
 *                valid, use fg_frame_device + fg_decode_frames_device
 * Asynchronous on `stream`.  FG_ERR_UNSUPPORTED (nothing launched): FG_RFC3164, or lines so long (average >= 768 bytes) that the
 * decoders stage heads only -- those keep the separate framing pass. */
int fg_frame_decode_device(fg_ctx* ctx, fg_format fmt, fg_framing framing, const uint8_t* d_bytes, uint64_t nbytes, int final,
                           uint64_t* d_offsets, uint64_t cap_frames, const fg_tables* tables, uint64_t avg_line_hint,
                           uint64_t* d_result, void* stream);

/* HOST-BUFFER decode: the batch is decoded into ctx-owned pinned host memory (`out` is filled with host pointers valid until
 * the next call on this ctx or fg_destroy).  Synchronous.  Entry-table capacity grows automatically.
 *   `bytes` AND `offsets` in PINNED memory (fg_alloc_pinned / hipHostRegister: where the batching framer accumulates lines),
 *   bytes 16-byte aligned: ZERO-COPY -- one launch, the kernels read the lines in place over the link and write the table columns
 *   straight into the pinned tables, both directions of the link busy at once.  The kernels load 16 bytes at a time: the pinned
 *   mapping must be READABLE UP TO nbytes ROUNDED UP TO 16 (fg_alloc_pinned memory is; a hipHostRegister'ed range must include
 *   those bytes), and bytes[0 .. that) and offsets[0 .. n] must each lie in ONE mapping -- buffers that do not qualify (the library
 *   checks the device view of the first and the last byte it will touch) silently take the sliced path below;
 *   anything else: the batch goes up as ~32 MiB slices on three streams (H2D, kernels, D2H of the tables), at the runtime's
 *   staged-copy speed for pageable memory. */
int fg_decode_batch(fg_ctx* ctx, fg_format fmt, const uint8_t* bytes, uint64_t nbytes,
                    const uint64_t* offsets, uint64_t n, fg_tables* out);

/* HOST-BUFFER RAW-STREAM decode: one call frames a chunk of the raw byte stream on the GPU
 * (FG_FRAME_LINE / FG_FRAME_NUL), validates UTF-8 per frame and decodes every frame -- what
 * LineSplitter::run / NulSplitter::run do per line (line_splitter.rs:17-54, nul_splitter.rs:18-60).
 *   final        nonzero: the stream ends with this chunk, so a trailing unterminated piece is a
 *                frame (BufRead semantics); zero: it is left for the next chunk
 *   out          tables in ctx-owned pinned host memory (as fg_decode_batch), *n_frames rows; a frame
 *                that is not valid UTF-8 has status FG_ST_BAD_UTF8
 *   out_offsets  ctx-owned host array, *n_frames + 1 entries: frame i = bytes[off[i] .. off[i+1])
 *                INCLUDING its terminator; spans in `out` are relative to off[i]
 *   consumed     bytes covered by the returned frames: carry bytes[consumed .. nbytes) over
 * Everything returned stays valid until the next call on this ctx. */
int fg_frame_decode_batch(fg_ctx* ctx, fg_format fmt, fg_framing framing, const uint8_t* bytes,
                          uint64_t nbytes, int final, fg_tables* out, const uint64_t** out_offsets,
                          uint64_t* n_frames, uint64_t* consumed);

/* Which form the last fg_decode_batch / fg_frame_decode_batch call on this ctx took (0 = none yet): the library picks by what the
 * caller's buffers allow, silently -- a harness that prices a leg against the link, or a test of one form, asks afterwards. */
enum {
    FG_PATH_DECODE_ZERO_COPY = 1,  /* fg_decode_batch: one launch, lines read over the link, tables written into pinned memory */
    FG_PATH_DECODE_SLICED = 2,     /* fg_decode_batch: hipMemcpy slices on three streams */
    FG_PATH_FRAME_FUSED = 3,       /* fg_frame_decode_batch: ONE launch -- the decode kernel frames the pinned chunk itself (round 6) */
    FG_PATH_FRAME_SLICED = 4,      /* fg_frame_decode_batch: upload slices + framing scan + decode per slice (rounds 3-5) */
    FG_PATH_FRAME_ONE_PIECE = 5    /* fg_frame_decode_batch: upload, frame, count on the host, decode */
};
int fg_last_host_path(const fg_ctx* ctx);

/* GELF ENCODER FROM THE TABLES: replaces GelfEncoder::encode (src/flowgger/encoder/
 * gelf_encoder.rs:59-115, serde_json 0.8 serialisation of the BTreeMap it builds) for a whole
 * decoded batch, without materialising Records: the JSON text of line i is written to
 * d_out[out_offsets[i] .. out_offsets[i+1]); a line whose decode failed (status != 0) produces
 * nothing.  `extra` = the output.gelf_extra table (may be NULL); the LTSV suffixes come from ctx.
 *   d_out_offsets  out, n + 1 entries (device)
 *   total          out (host): bytes needed; the call synchronises the stream to read it
 *   d_out == NULL  sizing call (only d_out_offsets / *total are produced);
 *   *total > out_cap -> FG_ERR_ENT_OVERFLOW, nothing written.
 * src_fmt says which decoder produced `tables`.  Same as fg_encode_device with FG_ENC_GELF / FG_MERGE_NONE. */
typedef struct fg_gelf_extra {
    uint32_t n;
    const char* const* keys;
    const char* const* values;
} fg_gelf_extra;
int fg_encode_gelf_device(fg_ctx* ctx, fg_format src_fmt, const uint8_t* d_bytes, uint64_t nbytes,
                          const uint64_t* d_offsets, uint64_t n, const fg_tables* tables,
                          const fg_gelf_extra* extra, uint8_t* d_out, uint64_t out_cap,
                          uint64_t* d_out_offsets, uint64_t* total, void* stream);

/* ANY ENCODER + MERGER FROM THE TABLES: replaces, for a whole decoded batch,
 *   Encoder::encode   encoder/{gelf,ltsv,rfc5424,rfc3164,passthrough}_encoder.rs   (trait: encoder/mod.rs:54-56)
 *   Merger::frame     merger/{line,nul,syslen}_merger.rs                             (trait: merger/mod.rs:30-32)
 * i.e. the `encoder.encode(decoded)?` of handle_line (splitter/line_splitter.rs:44-54) and the framing the outputs
 * apply to every message.  The encoded + framed bytes of line i are written to d_out[out_offsets[i] ..
 * out_offsets[i+1]); a line whose decode failed or whose encode returns Err produces nothing (the reference prints
 * the error and drops the line).  Records from all three decoders are accepted (GELF-sourced spans are JSON-unescaped
 * on the fly).  The capnp encoder is not provided.
 *   cfg->extra_*    output.gelf_extra (GELF) / output.ltsv_extra (LTSV) in the configuration table's iteration order
 *                   (a BTreeMap: sorted by key)
 *   cfg->prepend    RFC3164 / passthrough: the already formatted output.syslog_prepend_timestamp header (the
 *                   reference formats the wall clock per message, encoder/mod.rs:81-93); NULL = not configured
 *   cfg->now_ts     Record.ts of GELF records decoded without "timestamp" (rows flagged FG_F_TS_NOW; the reference
 *                   reads the wall clock at decode time, gelf_decoder.rs:109)
 *   d_out_offsets   out, n + 1 entries (device)
 *   d_enc_status    out, n bytes (device), may be NULL: 0 = encoded, 1 = the row's decode had failed,
 *                   else an encoder error (fg_encode_error_string)
 *   total           out (host): bytes needed; the call synchronises the stream to read it
 *   d_out == NULL   sizing call (only d_out_offsets / d_enc_status / *total are produced);
 *   *total > out_cap -> FG_ERR_ENT_OVERFLOW, nothing written. */
typedef enum fg_encoder { FG_ENC_GELF = 0, FG_ENC_LTSV = 1, FG_ENC_RFC5424 = 2, FG_ENC_RFC3164 = 3, FG_ENC_PASSTHROUGH = 4 } fg_encoder;
typedef enum fg_merger { FG_MERGE_NONE = 0, FG_MERGE_LINE = 1, FG_MERGE_NUL = 2, FG_MERGE_SYSLEN = 3 } fg_merger;
typedef struct fg_encode_cfg {
    fg_encoder encoder;
    fg_merger merger;
    uint32_t n_extra;
    const char* const* extra_keys;
    const char* const* extra_values;
    const char* prepend;
    double now_ts;
} fg_encode_cfg;
int fg_encode_device(fg_ctx* ctx, fg_format src_fmt, const fg_encode_cfg* cfg, const uint8_t* d_bytes, uint64_t nbytes,
                     const uint64_t* d_offsets, uint64_t n, const fg_tables* tables, uint8_t* d_out, uint64_t out_cap,
                     uint64_t* d_out_offsets, uint8_t* d_enc_status, uint64_t* total, void* stream);

/* The same WITHOUT a host synchronisation (a device-resident pipeline must not wait on the host for a byte count):
 * count, scan and write are queued on `stream` and the call returns.  Nothing comes back to the host:
 *   d_out_offsets[n]  (device) = the bytes the batch needs; when it exceeds out_cap the write kernel leaves d_out untouched --
 *                     the caller checks that word whenever it next synchronises (d_out_offsets / d_enc_status are always produced)
 *   ent_hint          an upper bound of the entries in `tables` (sizes the GELF encoder's key-ranking scratch instead of the
 *                     ent_used read-back): 0 = none (RFC5424 without structured data), ~0 = unknown
 * The first calls on a ctx may still synchronise while its scratch grows to the batch size (hipFree / hipMalloc).
 * replaces: the same `encoder.encode` + `merger.frame` as fg_encode_device (splitter/line_splitter.rs:44-54). */
int fg_encode_device_async(fg_ctx* ctx, fg_format src_fmt, const fg_encode_cfg* cfg, const uint8_t* d_bytes, uint64_t nbytes,
                           const uint64_t* d_offsets, uint64_t n, const fg_tables* tables, uint8_t* d_out, uint64_t out_cap,
                           uint64_t* d_out_offsets, uint8_t* d_enc_status, uint64_t ent_hint, void* stream);
/* The reference's exact &'static str of an encode status (0 / 1 -> "", unknown -> NULL). */
const char* fg_encode_error_string(uint8_t enc_status);

/* THE WHOLE handle_line OF A BATCH -- HOST BUFFERS IN, ENCODED STREAM OUT (BASELINE configs[0]: input -> decoder ->
 * encoder -> output).  Replaces, for every line of a batch, the body of handle_line (splitter/line_splitter.rs:44-54:
 * `decoder.decode(line)` -> `encoder.encode(decoded)` -> `tx.send(..)`) plus the Merger::frame the output applies
 * (merger/mod.rs:30-32), and -- when `framing` is FG_FRAME_LINE / FG_FRAME_NUL -- the splitter's own framing + UTF-8
 * check (line_splitter.rs:17-25, nul_splitter.rs:18-40).  Everything between the two PCIe copies stays in HBM: the
 * decode tables are never copied back; what returns is the encoded + framed byte stream in input order and one
 * verdict per line, i.e. what the output thread writes and what the splitter prints to stderr.
 *   framing == FG_FRAME_NONE: `bytes` / `offsets[n + 1]` are framed lines (as for fg_decode_batch); `final` is ignored.
 *   framing == FG_FRAME_LINE / _NUL: `bytes` is a raw stream chunk, `offsets` must be NULL and `n` is ignored; an
 *     unterminated tail is a frame only when `final` != 0, otherwise out->consumed < nbytes and the caller carries
 *     the rest over to the next call (as for fg_frame_decode_batch).
 * Results (pinned host memory owned by ctx, valid until the next host-buffer call on it):
 *   out / out_bytes / out_offsets[n + 1]   message i = out[out_offsets[i] .. out_offsets[i + 1]) -- empty when the
 *                                          line was dropped
 *   meta[n]         the table's meta column: meta & 0xFF = decoder status (fg_error_string; FG_ST_BAD_UTF8)
 *   enc_status[n]   0 = encoded, 1 = the decode had failed, else an encoder error (fg_encode_error_string)
 *   frame_offsets   the frames found in the chunk (n + 1 entries; NULL for FG_FRAME_NONE), for the stderr message
 *   n, consumed     lines handled, input bytes they cover */
typedef struct fg_transcoded {
    const uint8_t* out;
    uint64_t out_bytes;
    const uint64_t* out_offsets;
    const uint32_t* meta;
    const uint8_t* enc_status;
    const uint64_t* frame_offsets;
    uint64_t n;
    uint64_t consumed;
} fg_transcoded;
int fg_transcode_batch(fg_ctx* ctx, fg_format fmt, fg_framing framing, const fg_encode_cfg* cfg, const uint8_t* bytes,
                       uint64_t nbytes, const uint64_t* offsets, uint64_t n, int final, fg_transcoded* out);

/* Host memory for the framer's batch buffers (bytes, offsets): PAGE-LOCKED, through a process-wide pool -- a freed block is kept
 * (up to idle_bytes in total) and handed to the next request it fits, so a framer per connection does not pin and unpin megabytes per
 * connect; the bytes pinned through this allocator are capped at total_bytes, beyond which a request gets PAGEABLE memory (every
 * host-buffer entry point takes either: pinned = zero-copy / link-speed uploads, pageable = the runtime's staged copies).
 * Defaults: 8 GiB in total, 256 MiB idle (a measurement harness that keeps multi-GB batches pinned raises the cap: bench.py does).  Thread-safe.  fg_pinned_stats: what is pinned / idle / handed out right now. */
int fg_alloc_pinned(uint64_t bytes, void** out);
void fg_free_pinned(void* p);
int fg_set_pinned_limits(uint64_t total_bytes, uint64_t idle_bytes);
int fg_pinned_stats(uint64_t* pinned_bytes, uint64_t* idle_bytes, uint64_t* live_blocks);
/* 1: `p` (a pointer fg_alloc_pinned returned) is page-locked; 0: the cap was reached and the block is PAGEABLE memory -- the host-buffer
 * entry points then take their staged-copy forms, silently; -1: not a block of this allocator (fg_free_pinned ignores such a pointer:
 * it is the caller's to free). */
int fg_is_pinned(const void* p);

/* The host <-> device link of ctx's GPU, MEASURED: hipMemcpyAsync of a pinned buffer of `nbytes` (>= 64 MiB for a steady figure;
 * bench.py uses 1 GiB), best of three, in GB/s: gbps[0] host -> device, gbps[1] device -> host, gbps[2] both directions at once
 * on two streams (sum of both).  This is the roof the host-buffer entry points (fg_decode_batch, fg_transcode_batch ...) are
 * priced against -- SURVEY 8d: end-to-end is PCIe-bound -- instead of a nominal Gen5 figure. */
int fg_measure_link(fg_ctx* ctx, uint64_t nbytes, double gbps[3]);

/* CALIBRATION of the memory system the decode kernels run on, in the caller's process and on the caller's own resident buffer: a
 * plain streaming kernel over [d_src, d_src + nbytes) on `stream` (asynchronous; time it with events like a decode launch).
 *   FG_CALIB_COPY  float4 copy into d_dst (nbytes readable / writable, 16-byte aligned): 2 x nbytes of HBM traffic
 *   FG_CALIB_READ  read-only sweep (d_dst ignored): nbytes of traffic -- the roof of a decoder that writes little
 * bench.py reports both beside the decoder's achieved GB/s (roofline.copy_GBps / read_GBps) so that box-to-box variance is not
 * mistaken for a code change.  No reference analogue (measurement support). */
enum { FG_CALIB_COPY = 0, FG_CALIB_READ = 1,
       FG_CALIB_COPY_NT = 2,  /* the copy with non-temporal loads and stores */
       FG_CALIB_COPY_FLAT = 3 /* the copy as one 16-byte element per thread (no grid-stride loop) */ };
int fg_calibrate_device(fg_ctx* ctx, int mode, const uint8_t* d_src, uint8_t* d_dst, uint64_t nbytes, void* stream);

/* The reference's exact &'static str for a status code of a format (0 -> "", unknown -> NULL). */
const char* fg_error_string(fg_format fmt, uint8_t status);

/* Materialise rows [i0, i1) of HOST-visible tables into the canonical Record serialisation
 * (format documented in INTEGRATION.md; identical to what oracle/ emits for the reference
 * semantics).  out may be NULL to size; out_offsets (i1-i0+1 entries) may be NULL.
 * cfg supplies the LTSV suffixes (NULL otherwise).  Returns total bytes, or negative FG_ERR_*. */
int64_t fg_tables_serialize(fg_format fmt, const fg_cfg* cfg, const uint8_t* bytes,
                            const uint64_t* offsets, const fg_tables* tables, uint64_t i0,
                            uint64_t i1, uint8_t* out, uint64_t cap, uint64_t* out_offsets);

/* SIDE EFFECTS of decode(): the bytes the reference decoders write to the process's stdout while decoding rows [i0, i1), in row
 * order (SURVEY 8b "Side effects"; today one statement: LTSV's `println!("Missing value for name '{}'", name)`,
 * ltsv_decoder.rs:99 -- rows flagged FG_F_LTSV_NOVALUE).  A drop-in shim writes them to stdout after each batch.
 * `framing` says whether [offsets[i], offsets[i+1]) still carries its terminator (frames of fg_frame_decode_batch: FG_FRAME_LINE /
 * FG_FRAME_NUL) or is a bare line (FG_FRAME_NONE).  out may be NULL to size.  Returns the total bytes (even when > cap), or
 * negative FG_ERR_*.  Only `meta` and `hostname` of `tables` are read; hostname may be NULL (fg_transcode_batch returns the meta
 * column only): the count of a failed row then comes from its facility byte. */
int64_t fg_tables_stdout(fg_format fmt, fg_framing framing, const uint8_t* bytes, const uint64_t* offsets, const fg_tables* tables,
                         uint64_t i0, uint64_t i1, uint8_t* out, uint64_t cap);

/* Multi-GPU sharding plan (host): split n lines into g contiguous ranges balanced by BYTES;
 * line_starts receives g+1 line indices (line_starts[0] = 0, line_starts[g] = n). */
int fg_shard_plan(const uint64_t* offsets, uint64_t n, uint32_t g, uint64_t* line_starts);

/* ORDERED HOST GATHER of the multi-GPU path (order contract: handle_line runs in input order per
 * connection, src/flowgger/splitter/line_splitter.rs:17-54).  Every pointer is HOST memory (pinned or pageable);
 * plain memory moves threaded over row ranges, no GPU involved.
 *   fg_gather_size    sum of rows / used entries over the g parts (the capacity `out` needs)
 *   fg_gather_tables  parts = the tables of the g shards of ONE batch, in shard order (fg_shard_plan): `out` receives
 *                     their concatenation; ent_first is rebased onto the concatenated entry table, spans are
 *                     line-relative and are copied as they are.  out->n >= rows, out->ent_cap >= entries.
 *   fg_merge_tables   parts = sub-batches split off by FORMAT (BASELINE configuration 5: the reference has one decoder
 *                     per input, flowgger/mod.rs:413-422, so a mixed stream is decoded as tagged sub-batches);
 *                     index[k][j] = original position of row j of part k (strictly increasing inside a part, every
 *                     position 0..rows-1 exactly once): rows go back to their original positions;
 *                     src_part[i] (may be NULL) receives the part a row came from (= which decoder's error strings /
 *                     materialisation rules apply to it)
 *   fg_ordered_merge  the same for variable-size byte records (canonical Records, encoded messages): part k holds
 *                     m[k] records, record j = blobs[k][offs[k][j] .. offs[k][j+1]); out_offs receives rows+1 offsets;
 *                     returns the total bytes (call with out == NULL to size), negative FG_ERR_* on bad arguments */
int fg_gather_size(const fg_tables* parts, uint32_t g, uint64_t* n_rows, uint64_t* n_entries);
/* fg_merge_tables ON THE DEVICE (round 4): the parts' tables, index[k] and `out` are device memory (the sub-batches' tables as
 * fg_decode_batch_device left them, at most 8 parts); the rows go back to their arrival positions while everything is still in HBM --
 * ONE merged table then crosses the link instead of g tables plus a pass of the host's cores over all of them.  Round 5: the merged
 * ENTRIES are DENSE and in ARRIVAL order (row i's slice starts where row i-1's ends; ent_used = the sum of ent_count): the slots the
 * decoders' waves reserved but never used -- they lie between the slices of a decoder's own table -- do not cross the link.
 * Asynchronous on `stream`; the parts' entry counters are read on the device.  out->n must be the sum of the
 * parts' rows and out->ent_cap at least the sum of their ent_cap (checked); out->ent_used receives the merged entry count;
 * d_src_part (device, may be NULL) as src_part of fg_merge_tables.  The indices are the caller's contract (every position once). */
int fg_merge_tables_device(fg_ctx* ctx, const fg_tables* parts, uint32_t g, const uint64_t* const* d_index, const fg_tables* out,
                           uint8_t* d_src_part, void* stream);
int fg_gather_tables(const fg_tables* parts, uint32_t g, fg_tables* out);
int fg_merge_tables(const fg_tables* parts, uint32_t g, const uint64_t* const* index, fg_tables* out, uint8_t* src_part);
int64_t fg_ordered_merge(uint32_t g, const uint64_t* m, const uint64_t* const* index, const uint8_t* const* blobs,
                         const uint64_t* const* offs, uint8_t* out, uint64_t cap, uint64_t* out_offs);

/* Duration in milliseconds of the most recent decode kernel launch(es) of this ctx measured
 * with HIP events on the launch stream (0 when timing is disabled). fg_set_timing(ctx, 1)
 * enables event recording around every launch. */
int fg_set_timing(fg_ctx* ctx, int enabled);
int fg_last_kernel_ms(fg_ctx* ctx, float* ms);

/* Self-check of the dynamic chunk dispatch (round 5): the streaming decode kernels draw their chunks from a per-launch ticket counter
 * -- a word of a ctx-owned ring in device memory that is never reset: the host keeps what every word holds once the launches issued so
 * far have run (a launch adds exactly its number of chunks).  waits for the device, reads the ring back
 * and returns the number of words that differ from the host's books (0 = consistent; negative = an FG_ERR_* code).  A diagnostic
 * for tests and for a shim's debug builds; the decode path never calls it. */
int fg_ticket_ring_check(fg_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* FG_HIP_H */
