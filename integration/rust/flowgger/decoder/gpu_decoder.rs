//! gpu_decoder.rs -- the safe layer between flowgger's `Decoder` trait and libfg_hip (crate `fg-hip-sys`).
//!
//! Drop into `src/flowgger/decoder/` and add `mod gpu_decoder; pub use self::gpu_decoder::*;` to `decoder/mod.rs`.
//! What it provides, with the reference interfaces it keeps:
//!   * `GpuDecoder` implements `Decoder` (`decoder/mod.rs:44-46`): `decode(&self, &str) -> Result<Record, &'static str>`
//!     with the reference's exact error strings (`fg_error_string` returns the same `&'static str` tables);
//!   * `Clone + Send` (`decoder/mod.rs:29-36`, `input/tcp/tcp_input.rs:39-47`): one `fg_clone` per connection thread --
//!     a clone shares the device-side configuration and owns its stream and staging buffers; a ctx is NOT `Sync`;
//!   * `BatchDecoder::decode_batch`: N framed lines in ONE call (replaces the per-line loop of
//!     `splitter/line_splitter.rs:17,44-54`); `Record`s (`record.rs:70-82`) are materialised from the table rows with the
//!     same copy-time transformations the CPU decoders apply (`"_"` key prefix, `unescape_sd_value`, JSON unescape of
//!     flagged spans, LTSV suffix, RFC3164 message join) -- the Rust twin of `flowgger_amd/csrc/fg_materialize.cpp`.
//! There is no CPU fallback: without a gfx950 GPU `GpuDecoder::new` panics like any other mis-configured decoder.
//!
//! This file cannot be compiled in the build image (no rustc); `tests/test_rust_ffi_cpu.py` checks every `fg_*` item it
//! uses against `include/fg_hip.h` (through the generated `fg-hip-sys/src/lib.rs`).
use std::ffi::{CStr, CString};
use std::os::raw::{c_char, c_int};
use std::ptr;
use std::slice;
use std::str;

use fg_hip_sys::*;

use super::rfc5424_decoder::unescape_sd_value; // the reference's own function (rfc5424_decoder.rs:105-125), made `pub(crate)` by the patch
use super::Decoder;
use crate::flowgger::config::Config;
use crate::flowgger::record::{Record, SDValue, StructuredData};
use crate::flowgger::utils;

/// Which reference decoder a `GpuDecoder` stands in for.
#[derive(Clone, Copy, PartialEq, Eq, Debug)]
pub enum GpuFormat {
    Rfc5424, // decoder/rfc5424_decoder.rs:17-242
    Ltsv,    // decoder/ltsv_decoder.rs:23-267
    Gelf,    // decoder/gelf_decoder.rs:34-125
    Rfc3164, // decoder/rfc3164_decoder.rs:31-213
}

impl GpuFormat {
    pub(crate) fn raw(self) -> fg_format {
        match self {
            GpuFormat::Rfc5424 => FG_RFC5424,
            GpuFormat::Ltsv => FG_LTSV,
            GpuFormat::Gelf => FG_GELF,
            GpuFormat::Rfc3164 => FG_RFC3164,
        }
    }
}

/// `input.ltsv_schema` / `input.ltsv_suffixes` (ltsv_decoder.rs:24-84) in the shape `fg_cfg` wants; owned so that
/// clones can rebuild the C view.
#[derive(Clone, Default)]
pub struct LtsvSchema {
    pub names: Vec<CString>,
    pub types: Vec<u8>, // FG_T_STRING .. FG_T_U64
    pub suffix: [Option<CString>; 4], // bool, f64, i64, u64
}

pub struct GpuDecoder {
    ctx: *mut fg_ctx,
    fmt: GpuFormat,
    ltsv: LtsvSchema,
}

// One ctx per thread (`Send`), never shared (`!Sync` by the raw pointer): exactly the reference's contract.
unsafe impl Send for GpuDecoder {}

impl GpuDecoder {
    pub fn rfc5424(_config: &Config, device: i32) -> GpuDecoder {
        GpuDecoder::create(GpuFormat::Rfc5424, LtsvSchema::default(), device)
    }
    pub fn gelf(_config: &Config, device: i32) -> GpuDecoder {
        GpuDecoder::create(GpuFormat::Gelf, LtsvSchema::default(), device)
    }
    /// `LTSVDecoder::new` (ltsv_decoder.rs:24-84): the caller parses the two TOML tables exactly as the reference does
    /// (same panics for unsupported types) and hands the result over.
    pub fn ltsv(schema: LtsvSchema, device: i32) -> GpuDecoder {
        GpuDecoder::create(GpuFormat::Ltsv, schema, device)
    }
    /// `RFC3164Decoder::new` (rfc3164_decoder.rs:22-28).  The reference reads the wall-clock year per parse (:179):
    /// FG_YEAR_NOW makes the library do the same at every decode call.  `tz` = the zone table behind
    /// `time_tz::timezones::get_by_name` (:195), see `TzTable`.
    pub fn rfc3164(_config: &Config, tz: &TzTable, device: i32) -> GpuDecoder {
        let d = GpuDecoder::create(GpuFormat::Rfc3164, LtsvSchema::default(), device);
        let names: Vec<*const c_char> = tz.names.iter().map(|n| n.as_ptr()).collect();
        let table = fg_tz_table {
            n_zones: tz.names.len() as u32,
            names: names.as_ptr(),
            zone_first: tz.zone_first.as_ptr(),
            utc_start: tz.utc_start.as_ptr(),
            utc_offset: tz.utc_offset.as_ptr(),
        };
        let cfg = fg_rfc3164_cfg { current_year: FG_YEAR_NOW, tz: &table };
        let rc = unsafe { fg_set_rfc3164(d.ctx, &cfg) };
        assert_eq!(rc, FG_OK, "fg_set_rfc3164 failed: {}", rc);
        d
    }

    fn create(fmt: GpuFormat, ltsv: LtsvSchema, device: i32) -> GpuDecoder {
        let mut ctx: *mut fg_ctx = ptr::null_mut();
        let rc = with_cfg(&ltsv, |cfg| unsafe { fg_create(device as c_int, cfg, &mut ctx) });
        if rc == FG_ERR_NO_DEVICE {
            panic!("libfg_hip: no gfx950 GPU is usable and there is no CPU fallback");
        }
        assert_eq!(rc, FG_OK, "fg_create failed: {}", rc);
        assert_eq!(unsafe { fg_abi_version() }, FG_ABI_VERSION);
        GpuDecoder { ctx, fmt, ltsv }
    }

    pub fn format(&self) -> GpuFormat {
        self.fmt
    }
    pub(crate) fn raw_ctx(&self) -> *mut fg_ctx {
        self.ctx
    }
    pub(crate) fn ltsv(&self) -> &LtsvSchema {
        &self.ltsv
    }
}

/// The IANA zone table of `fg_tz_table`: names sorted bytewise; zone i owns entries
/// `[zone_first[i], zone_first[i + 1])` of `(utc_start, utc_offset)`; a zone's first entry starts at `i64::MIN`.
pub struct TzTable {
    pub names: Vec<CString>,
    pub zone_first: Vec<u32>,
    pub utc_start: Vec<i64>,
    pub utc_offset: Vec<i32>,
}

fn with_cfg<R>(s: &LtsvSchema, f: impl FnOnce(*const fg_cfg) -> R) -> R {
    if s.names.is_empty() && s.suffix.iter().all(|x| x.is_none()) {
        return f(ptr::null());
    }
    let names: Vec<*const c_char> = s.names.iter().map(|n| n.as_ptr()).collect();
    let suf = |k: usize| s.suffix[k].as_ref().map_or(ptr::null(), |c| c.as_ptr());
    let cfg = fg_cfg {
        n_schema: names.len() as u32,
        schema_names: names.as_ptr(),
        schema_types: s.types.as_ptr(),
        suffix_bool: suf(0),
        suffix_f64: suf(1),
        suffix_i64: suf(2),
        suffix_u64: suf(3),
    };
    f(&cfg)
}

impl Clone for GpuDecoder {
    /// `clone_boxed` (decoder/mod.rs:29-36): cheap -- the clone shares the device-side configuration.
    fn clone(&self) -> GpuDecoder {
        let mut ctx: *mut fg_ctx = ptr::null_mut();
        let rc = unsafe { fg_clone(self.ctx, &mut ctx) };
        assert_eq!(rc, FG_OK, "fg_clone failed: {}", rc);
        GpuDecoder { ctx, fmt: self.fmt, ltsv: self.ltsv.clone() }
    }
}

impl Drop for GpuDecoder {
    fn drop(&mut self) {
        unsafe { fg_destroy(self.ctx) }
    }
}

/// The batch entry point the batching framer calls (one call per accumulated batch).
pub trait BatchDecoder: Decoder {
    /// `bytes` holds the framed lines back to back; line i = `bytes[offsets[i]..offsets[i + 1]]` (valid UTF-8, framing
    /// bytes stripped as `BufRead::lines()` / `split(0)` / syslen do).  `bytes` must be readable up to its length rounded
    /// up to 16 (keep 16 spare bytes of capacity).  Results are in input order.
    fn decode_batch(&self, bytes: &[u8], offsets: &[u64]) -> Vec<Result<Record, &'static str>>;
}

impl Decoder for GpuDecoder {
    /// The reference's trait method: a batch of one -- one launch, one stream synchronisation and one table copy per record
    /// (tens of microseconds; tools/host_path_bench.py --workload latency).  The per-record callers (`udp_input.rs:139`,
    /// `redis_input.rs:159`, `file/worker.rs:116`) should go through `MicroBatcher` below instead.
    fn decode(&self, line: &str) -> Result<Record, &'static str> {
        let mut buf = Vec::with_capacity(line.len() + 16);
        buf.extend_from_slice(line.as_bytes());
        let offsets = [0u64, line.len() as u64];
        self.decode_batch(&buf, &offsets).pop().unwrap()
    }
    /// `Decoder::as_gpu` is a PROVIDED method the patch adds to the trait (`decoder/mod.rs:44-46`: `fn as_gpu(&self) ->
    /// Option<&GpuDecoder> { None }`); only this impl overrides it.  It is how `GpuSplitter::run`, which receives a
    /// `Box<dyn Decoder>` like every splitter (`splitter/mod.rs:18-26`), gets at the ctx.
    fn as_gpu(&self) -> Option<&GpuDecoder> {
        Some(self)
    }
}

/// The per-record callers' adapter (`input/udp_input.rs:78-88`, `input/redis_input.rs:150-165`, `input/file/worker.rs:110-120`
/// call `decoder.decode(record)` once per record): records are parked for at most `max_latency` (or until `max_lines` are there)
/// and decoded in ONE GPU call; results come back in arrival order through `deliver`.  The UDP loop becomes
/// ```ignore
/// socket.set_read_timeout(mb.wait())?;                 // None = block: nothing is parked
/// match socket.recv_from(&mut buf) { Ok((n, _)) => mb.push(&buf[..n]), Err(_) => {} }
/// mb.poll(|res, rec| match res.and_then(|r| encoder.encode(r)) { Ok(b) => tx.send(b).unwrap(), Err(e) => { let _ = writeln!(stderr(), "{}", e); } });
/// ```
pub struct MicroBatcher<'a> {
    decoder: &'a GpuDecoder,
    bytes: Vec<u8>,
    offsets: Vec<u64>,
    first: Option<std::time::Instant>,
    max_lines: usize,
    max_latency: std::time::Duration,
}

impl<'a> MicroBatcher<'a> {
    pub fn new(decoder: &'a GpuDecoder, max_lines: usize, max_latency: std::time::Duration) -> MicroBatcher<'a> {
        MicroBatcher { decoder, bytes: Vec::new(), offsets: vec![0], first: None, max_lines, max_latency }
    }
    /// One record (`handle_record`, `udp_input.rs:125-143`); invalid UTF-8 is reported at its place by `poll` / `flush`.
    pub fn push(&mut self, record: &[u8]) {
        if self.first.is_none() {
            self.first = Some(std::time::Instant::now());
        }
        self.bytes.extend_from_slice(record);
        self.offsets.push(self.bytes.len() as u64);
    }
    /// How long the caller may block in its receive call before `poll` is due; `None` = nothing is parked.
    pub fn wait(&self) -> Option<std::time::Duration> {
        self.first.map(|t| self.max_latency.checked_sub(t.elapsed()).unwrap_or(std::time::Duration::from_millis(1)))
    }
    /// Decodes what is parked when it is due (full, or the oldest record has waited `max_latency`).
    pub fn poll<F: FnMut(Result<Record, &'static str>, &[u8])>(&mut self, deliver: F) {
        let due = self.offsets.len() - 1 >= self.max_lines || self.first.map_or(false, |t| t.elapsed() >= self.max_latency);
        if due {
            self.flush(deliver);
        }
    }
    pub fn flush<F: FnMut(Result<Record, &'static str>, &[u8])>(&mut self, mut deliver: F) {
        let n = self.offsets.len() - 1;
        if n == 0 {
            return;
        }
        // "Invalid UTF-8 input" (udp_input.rs:135-138) never reaches decode(): such records are cut out of the batch
        let valid: Vec<bool> = (0..n).map(|i| str::from_utf8(&self.bytes[self.offsets[i] as usize..self.offsets[i + 1] as usize]).is_ok()).collect();
        let mut ok_bytes = Vec::with_capacity(self.bytes.len() + 16);
        let mut ok_offs = vec![0u64];
        for i in 0..n {
            if valid[i] {
                ok_bytes.extend_from_slice(&self.bytes[self.offsets[i] as usize..self.offsets[i + 1] as usize]);
                ok_offs.push(ok_bytes.len() as u64);
            }
        }
        let mut res = self.decoder.decode_batch(&ok_bytes, &ok_offs).into_iter();
        for i in 0..n {
            let rec = &self.bytes[self.offsets[i] as usize..self.offsets[i + 1] as usize];
            if valid[i] {
                deliver(res.next().unwrap(), rec);
            } else {
                deliver(Err("Invalid UTF-8 input"), rec);
            }
        }
        self.bytes.clear();
        self.offsets.truncate(1);
        self.first = None;
    }
}

impl BatchDecoder for GpuDecoder {
    fn decode_batch(&self, bytes: &[u8], offsets: &[u64]) -> Vec<Result<Record, &'static str>> {
        assert!(!offsets.is_empty());
        let n = (offsets.len() - 1) as u64;
        let mut t: fg_tables = unsafe { std::mem::zeroed() };
        let rc = unsafe { fg_decode_batch(self.ctx, self.fmt.raw(), bytes.as_ptr(), bytes.len() as u64, offsets.as_ptr(), n, &mut t) };
        assert_eq!(rc, FG_OK, "fg_decode_batch failed: {} (hip error {})", rc, unsafe { fg_last_hip_error(self.ctx) });
        // `t` points into pinned memory owned by the ctx: valid until the next call on this ctx
        (0..n as usize).map(|i| unsafe { materialise(self.fmt, &self.ltsv, &t, bytes, offsets[i] as usize, i) }).collect()
    }
}

/// `status` -> the reference's `&'static str` (the table lives in the library's read-only data).
pub fn error_str(fmt: GpuFormat, status: u8) -> &'static str {
    let p = unsafe { fg_error_string(fmt.raw(), status) };
    if p.is_null() {
        return "unknown decoder status";
    }
    unsafe { str::from_utf8_unchecked(CStr::from_ptr(p).to_bytes()) }
}

/// One table row + the line's bytes -> the `Record` the CPU decoder would have returned.
///
/// # Safety
/// `t` must be the tables of a decode of `bytes` and `i < t.n`; `line0` = the line's first byte in `bytes`.
pub unsafe fn materialise(fmt: GpuFormat, ltsv: &LtsvSchema, t: &fg_tables, bytes: &[u8], line0: usize, i: usize) -> Result<Record, &'static str> {
    let meta = *t.meta.add(i);
    let status = FG_META_STATUS(meta);
    if status != 0 {
        return Err(error_str(fmt, status));
    }
    let flags = FG_META_FLAGS(meta) as c_int;
    let retry = flags & FG_F_GELF_RETRY != 0;
    let line = &bytes[line0..];
    let raw = |sp: fg_span| -> &[u8] { &line[sp.off as usize..(sp.off + sp.len) as usize] };
    let plain = |sp: fg_span| -> String { str::from_utf8_unchecked(raw(sp)).to_owned() };
    let opt = |sp: fg_span, esc: bool| -> Option<String> {
        if sp.len == FG_NONE {
            None
        } else if esc {
            Some(json_unescape(raw(sp), retry))
        } else {
            Some(plain(sp))
        }
    };
    let gelf = fmt == GpuFormat::Gelf;
    let hostname = opt(*t.hostname.add(i), gelf && flags & FG_F_HOST_ESC != 0).unwrap_or_default();
    let msg_span = *t.msg.add(i);
    let msg = if fmt == GpuFormat::Rfc3164 && flags & FG_F_MSG_JOIN != 0 && msg_span.len != FG_NONE {
        // `_log_tokens[1..].join(" ")` (rfc3164_decoder.rs:70)
        Some(plain(msg_span).split_whitespace().collect::<Vec<_>>().join(" "))
    } else {
        opt(msg_span, gelf && flags & FG_F_MSG_ESC != 0)
    };
    let full_msg = opt(*t.full_msg.add(i), gelf && flags & FG_F_FULLMSG_ESC != 0);

    // structured data: the line's slice of the entry table, in decoder order
    let (first, cnt) = (*t.ent_first.add(i) as usize, *t.ent_count.add(i) as usize);
    let mut sd_vec: Vec<StructuredData> = Vec::new();
    if cnt != 0 && fmt != GpuFormat::Rfc5424 {
        sd_vec.push(StructuredData::new(None)); // ltsv_decoder.rs:88,215; gelf_decoder.rs:35,119
    }
    for e in first..first + cnt {
        let name = *t.ent_name.add(e);
        let ty = *t.ent_type.add(e) as c_int;
        let ef = *t.ent_flags.add(e) as c_int;
        if ty == FG_T_SDID {
            sd_vec.push(StructuredData::new(Some(&plain(name))));
            continue;
        }
        let mut key = if ef & FG_EF_NAME_ESC != 0 { json_unescape(raw(name), retry) } else { plain(name) };
        if !gelf || !key.starts_with('_') {
            key.insert(0, '_'); // rfc5424_decoder.rs:220-227, ltsv_decoder.rs:128-135; GELF only when missing, gelf_decoder.rs:99-103
        }
        if ef & FG_EF_SUFFIX != 0 && ty >= FG_T_BOOL && ty <= FG_T_U64 {
            if let Some(s) = &ltsv.suffix[(ty - FG_T_BOOL) as usize] {
                key.push_str(s.to_str().unwrap()); // ltsv_decoder.rs:131-136
            }
        }
        let v = *t.ent_val.add(e);
        let value = match ty {
            x if x == FG_T_STRING => {
                let (vo, vl) = (v as u32 as usize, (v >> 32) as usize);
                let bytes = &line[vo..vo + vl];
                if ef & FG_EF_VAL_ESC == 0 {
                    SDValue::String(str::from_utf8_unchecked(bytes).to_owned())
                } else if fmt == GpuFormat::Rfc5424 {
                    SDValue::String(unescape_sd_value(str::from_utf8_unchecked(bytes)))
                } else {
                    SDValue::String(json_unescape(bytes, retry))
                }
            }
            x if x == FG_T_BOOL => SDValue::Bool(v != 0),
            x if x == FG_T_F64 => SDValue::F64(f64::from_bits(v)),
            x if x == FG_T_I64 => SDValue::I64(v as i64),
            x if x == FG_T_U64 => SDValue::U64(v),
            _ => SDValue::Null,
        };
        sd_vec.last_mut().expect("an RFC5424 entry slice starts with its FG_T_SDID header").pairs.push((key, value));
    }
    let ts = if flags & FG_F_TS_NOW != 0 {
        utils::PreciseTimestamp::now().as_f64() // gelf_decoder.rs:109
    } else {
        *t.ts.add(i)
    };
    let fac = FG_META_FACILITY(meta);
    let sev = FG_META_SEVERITY(meta);
    Ok(Record {
        ts,
        hostname,
        facility: if fac == 0xFF { None } else { Some(fac) },
        severity: if sev == 0xFF { None } else { Some(sev) },
        appname: opt(*t.appname.add(i), false),
        procid: opt(*t.procid.add(i), false),
        msgid: opt(*t.msgid.add(i), false),
        msg,
        full_msg,
        sd: if sd_vec.is_empty() { None } else { Some(sd_vec) },
    })
}

/// serde_json 0.8's `parse_escape` over an ALREADY VALIDATED string body (the kernel rejected malformed escapes).
/// `retry`: the line was accepted through `line.replace('\n', "\\n")` (gelf_decoder.rs:44-46) -- the kernels apply that
/// replace on the fly, so a backslash followed by a raw LF in the span stands for an escaped backslash and an `n`.
fn json_unescape(p: &[u8], retry: bool) -> String {
    let mut out: Vec<u8> = Vec::with_capacity(p.len());
    let hex = |c: u8| -> u32 {
        match c {
            b'0'..=b'9' => (c - b'0') as u32,
            b'a'..=b'f' => (c - b'a' + 10) as u32,
            b'A'..=b'F' => (c - b'A' + 10) as u32,
            _ => 0,
        }
    };
    let mut i = 0;
    while i < p.len() {
        let c = p[i];
        if c != b'\\' || i + 1 >= p.len() {
            out.push(c);
            i += 1;
            continue;
        }
        let e = p[i + 1];
        i += 2;
        if retry && e == b'\n' {
            out.extend_from_slice(b"\\n");
            continue;
        }
        match e {
            b'b' => out.push(0x08),
            b'f' => out.push(0x0c),
            b'n' => out.push(b'\n'),
            b'r' => out.push(b'\r'),
            b't' => out.push(b'\t'),
            b'u' => {
                if i + 4 > p.len() {
                    break;
                }
                let mut n1 = hex(p[i]) << 12 | hex(p[i + 1]) << 8 | hex(p[i + 2]) << 4 | hex(p[i + 3]);
                i += 4;
                if (0xD800..=0xDBFF).contains(&n1) && i + 6 <= p.len() {
                    let n2 = hex(p[i + 2]) << 12 | hex(p[i + 3]) << 8 | hex(p[i + 4]) << 4 | hex(p[i + 5]);
                    i += 6;
                    n1 = (((n1 - 0xD800) << 10) | (n2 - 0xDC00)) + 0x10000;
                }
                let mut b = [0u8; 4];
                out.extend_from_slice(std::char::from_u32(n1).unwrap_or('\u{FFFD}').encode_utf8(&mut b).as_bytes());
            }
            _ => out.push(e), // " \ /
        }
    }
    unsafe { String::from_utf8_unchecked(out) }
}

/// View of ctx-owned host arrays (valid until the next call on the ctx).
pub(crate) unsafe fn host_slice<'a, T>(p: *const T, n: u64) -> &'a [T] {
    if p.is_null() || n == 0 {
        &[]
    } else {
        slice::from_raw_parts(p, n as usize)
    }
}
