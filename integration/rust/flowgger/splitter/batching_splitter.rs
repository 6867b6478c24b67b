//! batching_splitter.rs -- `LineSplitter` / `NulSplitter` with ONE GPU call per chunk of the stream instead of one
//! `decoder.decode()` per line.  Drop into `src/flowgger/splitter/` (replaces the `run` bodies of
//! `line_splitter.rs:10-41` and `nul_splitter.rs:10-60`; `handle_line`, `line_splitter.rs:44-54`, becomes the loop over a
//! batch's results).  Two variants:
//!   * `run_decode`    raw chunk -> `fg_frame_decode_batch` (GPU framing + UTF-8 validation + decode); `Record`s are
//!                     materialised on the host and go through the unchanged `Encoder` trait objects;
//!   * `run_transcode` raw chunk -> `fg_transcode_batch` (framing + decode + encode + merger on the GPU): only the
//!                     encoded, already framed bytes come back -- for the encoders libfg_hip provides (all but capnp).
//! Per-connection order and the reference's stderr messages are preserved.
use std::io::{stderr, BufReader, Read, Write};
use std::ptr;
use std::sync::mpsc::SyncSender;

use fg_hip_sys::*;

use crate::flowgger::decoder::gpu_decoder::{error_str, host_slice, materialise, GpuDecoder, LtsvSchema};
use crate::flowgger::encoder::Encoder;

const CHUNK: usize = 8 << 20;

pub struct GpuSplitter {
    pub framing: fg_framing, // FG_FRAME_LINE (lines()) or FG_FRAME_NUL (split(0))
}

impl GpuSplitter {
    pub fn run_decode<T: Read>(&self, mut reader: BufReader<T>, tx: SyncSender<Vec<u8>>, decoder: GpuDecoder, encoder: Box<dyn Encoder>, ltsv: &LtsvSchema) {
        let mut buf: Vec<u8> = Vec::with_capacity(2 * CHUNK);
        let mut eof = false;
        while !eof || !buf.is_empty() {
            eof = eof || fill(&mut reader, &mut buf, CHUNK) == 0;
            if buf.is_empty() {
                break;
            }
            let nbytes = buf.len();
            buf.resize(nbytes + 16, 0); // readable slack
            let mut t: fg_tables = unsafe { std::mem::zeroed() };
            let (mut off, mut n, mut used) = (ptr::null(), 0u64, 0u64);
            let rc = unsafe {
                fg_frame_decode_batch(decoder.raw_ctx(), fmt_of(&decoder), self.framing, buf.as_ptr(), nbytes as u64, eof as i32, &mut t, &mut off, &mut n, &mut used)
            };
            assert_eq!(rc, FG_OK, "fg_frame_decode_batch failed: {}", rc);
            let offs = unsafe { host_slice(off, n + 1) };
            for i in 0..n as usize {
                let frame = &buf[offs[i] as usize..offs[i + 1] as usize];
                let status = FG_META_STATUS(unsafe { *t.meta.add(i) });
                if status == FG_ST_BAD_UTF8 {
                    let _ = writeln!(stderr(), "Invalid UTF-8 input"); // line_splitter.rs:22-25, nul_splitter.rs:35-38
                    continue;
                }
                let res = unsafe { materialise(decoder.format(), ltsv, &t, &buf, offs[i] as usize, i) }.and_then(|r| encoder.encode(r));
                match res {
                    Ok(bytes) => tx.send(bytes).unwrap(), // line_splitter.rs:52
                    Err(e) => {
                        let text = String::from_utf8_lossy(frame);
                        let text = text.trim();
                        if self.framing == FG_FRAME_NUL && text.is_empty() {
                            continue; // nul_splitter.rs:41-46
                        }
                        let _ = writeln!(stderr(), "{}: [{}]", e, text); // line_splitter.rs:37-39
                    }
                }
            }
            buf.truncate(nbytes);
            if used == 0 && n == 0 && !eof {
                continue; // one frame longer than the chunk: read more
            }
            buf.drain(..used as usize); // an unterminated tail waits for more bytes
        }
    }

    pub fn run_transcode<T: Read>(&self, mut reader: BufReader<T>, tx: SyncSender<Vec<u8>>, decoder: GpuDecoder, enc: &fg_encode_cfg) {
        let mut buf: Vec<u8> = Vec::with_capacity(2 * CHUNK);
        let mut eof = false;
        while !eof || !buf.is_empty() {
            eof = eof || fill(&mut reader, &mut buf, CHUNK) == 0;
            if buf.is_empty() {
                break;
            }
            let nbytes = buf.len();
            buf.resize(nbytes + 16, 0);
            let mut r: fg_transcoded = unsafe { std::mem::zeroed() };
            let rc = unsafe {
                fg_transcode_batch(decoder.raw_ctx(), fmt_of(&decoder), self.framing, enc, buf.as_ptr(), nbytes as u64, ptr::null(), 0, eof as i32, &mut r)
            };
            assert_eq!(rc, FG_OK, "fg_transcode_batch failed: {}", rc);
            let (out, offs, meta, es, frames) = unsafe {
                (host_slice(r.out, r.out_bytes), host_slice(r.out_offsets, r.n + 1), host_slice(r.meta, r.n), host_slice(r.enc_status, r.n), host_slice(r.frame_offsets, r.n + 1))
            };
            for i in 0..r.n as usize {
                let status = FG_META_STATUS(meta[i]);
                let text = || String::from_utf8_lossy(&buf[frames[i] as usize..frames[i + 1] as usize]).trim().to_owned();
                if status == FG_ST_BAD_UTF8 {
                    let _ = writeln!(stderr(), "Invalid UTF-8 input");
                } else if status != 0 {
                    let t = text();
                    if !(self.framing == FG_FRAME_NUL && t.is_empty()) {
                        let _ = writeln!(stderr(), "{}: [{}]", error_str(decoder.format(), status), t);
                    }
                } else if es[i] > 1 {
                    let e = unsafe { std::ffi::CStr::from_ptr(fg_encode_error_string(es[i])) }.to_string_lossy();
                    let _ = writeln!(stderr(), "{}: [{}]", e, text());
                } else {
                    // one message per send keeps the outputs' queue semantics (line_splitter.rs:52); the messages are
                    // already framed by the merger, so an output that writes the queue back to back may also take
                    // `out` whole
                    tx.send(out[offs[i] as usize..offs[i + 1] as usize].to_vec()).unwrap();
                }
            }
            buf.truncate(nbytes);
            if r.consumed == 0 && r.n == 0 && !eof {
                continue;
            }
            buf.drain(..r.consumed as usize);
        }
    }
}

fn fmt_of(d: &GpuDecoder) -> fg_format {
    use crate::flowgger::decoder::gpu_decoder::GpuFormat::*;
    match d.format() {
        Rfc5424 => FG_RFC5424,
        Ltsv => FG_LTSV,
        Gelf => FG_GELF,
        Rfc3164 => FG_RFC3164,
    }
}

fn fill<T: Read>(reader: &mut BufReader<T>, buf: &mut Vec<u8>, want: usize) -> usize {
    let have = buf.len();
    buf.resize(have + want, 0);
    let mut got = 0;
    while got < want {
        match reader.read(&mut buf[have + got..]) {
            Ok(0) => break,
            Ok(k) => got += k,
            Err(ref e) if e.kind() == std::io::ErrorKind::Interrupted => continue,
            Err(_) => break,
        }
    }
    buf.truncate(have + got);
    got
}
