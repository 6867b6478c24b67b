//! batching_splitter.rs -- `LineSplitter` / `NulSplitter` with ONE GPU call per batch of the stream instead of one
//! `decoder.decode()` per line.  Drop into `src/flowgger/splitter/` beside `line_splitter.rs`; `GpuSplitter` implements the
//! reference's `Splitter<T>` (`splitter/mod.rs:18-26`), so an input selects it exactly like the others
//! (`stdin_input.rs:57-64`, `tcp_input.rs:77-84`: one more arm in the `match &config.framing`).  Two bodies:
//!   * `run_decode`    raw chunk -> `fg_frame_decode_batch` (GPU framing + UTF-8 validation + decode); `Record`s are
//!                     materialised on the host and go through the unchanged `Encoder` trait object (what `run` uses);
//!   * `run_transcode` raw chunk -> `fg_transcode_batch` (framing + decode + encode + merger on the GPU): only the
//!                     encoded, already framed bytes come back -- for the encoders libfg_hip provides (all but capnp).
//! WHEN a batch goes to the GPU (the reference handles every line the moment `lines()` yields it, `line_splitter.rs:17`): when
//! `max_bytes` have accumulated, or -- the usual case on a live connection -- when a read returns LESS than it could hold: the
//! peer has nothing more in flight right now, so what has arrived is decoded now instead of waiting for a full chunk.  A read
//! timeout (`ErrorKind::WouldBlock`, `input.timeout` via `tcp_input.rs:41`) decodes the complete lines that are buffered, prints
//! the reference's message and ends the connection (`line_splitter.rs:26-33`, `nul_splitter.rs:22-29`); the unterminated tail is
//! dropped exactly as `lines()` drops it.  Per-connection order and the reference's stderr / stdout texts are preserved.
use std::io::{stderr, stdout, BufRead, BufReader, ErrorKind, Read, Write};
use std::ptr;
use std::sync::mpsc::SyncSender;

use fg_hip_sys::*;

use super::Splitter;
use crate::flowgger::decoder::gpu_decoder::{error_str, host_slice, materialise, GpuDecoder};
use crate::flowgger::decoder::Decoder;
use crate::flowgger::encoder::Encoder;

/// A batch never grows beyond this (the GPU path runs at link speed long before).
const MAX_BYTES: usize = 8 << 20;
/// Capacity of the reader the splitter reads through.  `fill()` never blocks while complete frames are buffered, so under sustained
/// load a batch is what ONE read returns: with the 8 KiB `BufReader` the reference's inputs build (`tcp_input.rs:77`,
/// `stdin_input.rs:57`) that is ~32 lines per GPU call, each paying the call's fixed cost.  The splitter therefore
/// re-wraps the source: whatever the input's reader has buffered is taken over, the source itself moves into a reader of this size.
const READER_CAPACITY: usize = 1 << 20;

/// The input's `BufReader` (8 KiB by default) -> one of READER_CAPACITY over the same source; bytes it had already buffered go
/// to `carry` (they are the first bytes of the first batch).  A reader that is already large enough is kept.
fn widen<T: Read>(reader: BufReader<T>, carry: &mut Vec<u8>) -> BufReader<T> {
    if reader.capacity() >= READER_CAPACITY {
        return reader;
    }
    carry.extend_from_slice(reader.buffer());
    BufReader::with_capacity(READER_CAPACITY, reader.into_inner())
}

pub struct GpuSplitter {
    pub framing: fg_framing, // FG_FRAME_LINE (lines()) or FG_FRAME_NUL (split(0))
}

/// How a `fill` ended.
#[derive(PartialEq, Eq, Clone, Copy)]
enum Fill {
    Data,  // bytes were added and the source has nothing more right now (or the batch is full)
    Eof,   // end of the stream (bytes may have been added before it)
    Idle,  // ErrorKind::WouldBlock: the read timeout expired
    Error, // any other error: the reference's `_ => return`
}

impl<T: Read> Splitter<T> for GpuSplitter {
    /// The reference's signature (`splitter/mod.rs:18-26`).  The boxed decoder is a `GpuDecoder` when the configuration
    /// selected one (`Decoder::as_gpu`, a provided trait method that only `GpuDecoder` overrides -- INTEGRATION.md section 3);
    /// with any other decoder this splitter has nothing to batch for.
    fn run(&self, buf_reader: BufReader<T>, tx: SyncSender<Vec<u8>>, decoder: Box<dyn Decoder>, encoder: Box<dyn Encoder>) {
        let gpu = decoder.as_gpu().expect("framing = \"gpu-line\" / \"gpu-nul\" needs a GPU decoder (input.format + input.gpu = true)");
        self.run_decode(buf_reader, tx, gpu, encoder)
    }
}

impl GpuSplitter {
    pub fn run_decode<T: Read>(&self, reader: BufReader<T>, tx: SyncSender<Vec<u8>>, decoder: &GpuDecoder, encoder: Box<dyn Encoder>) {
        let mut buf: Vec<u8> = Vec::with_capacity(2 * MAX_BYTES);
        let mut reader = widen(reader, &mut buf);
        // does `buf` hold a complete frame?  Kept as a flag: what a decode leaves behind is an unterminated tail (no frame), and
        // `fill` looks only at the bytes it adds -- the buffer is never rescanned (a long unterminated frame was quadratic)
        let mut framed = has_frame(&buf, self.framing);
        // what the input's reader had buffered may already hold complete frames: they are decoded BEFORE the source is read again -- `fill`
        // blocks in `fill_buf` up to the socket's read timeout, and nothing decodable may wait behind that
        let mut carried = framed;
        loop {
            let how = if carried { Fill::Data } else { fill(&mut reader, &mut buf, MAX_BYTES, self.framing, &mut framed) };
            carried = false;
            let last = how != Fill::Data;
            let eof = how == Fill::Eof;
            if !buf.is_empty() && (eof || framed) {
                let nbytes = buf.len();
                buf.resize(nbytes + 16, 0); // readable slack
                let mut t: fg_tables = unsafe { std::mem::zeroed() };
                let (mut off, mut n, mut used) = (ptr::null(), 0u64, 0u64);
                let rc = unsafe {
                    fg_frame_decode_batch(decoder.raw_ctx(), decoder.format().raw(), self.framing, buf.as_ptr(), nbytes as u64, eof as i32, &mut t, &mut off, &mut n, &mut used)
                };
                assert_eq!(rc, FG_OK, "fg_frame_decode_batch failed: {}", rc);
                let offs = unsafe { host_slice(off, n + 1) };
                print_side_effects(decoder, self.framing, &buf, offs, &t, n);
                for i in 0..n as usize {
                    let frame = &buf[offs[i] as usize..offs[i + 1] as usize];
                    let status = FG_META_STATUS(unsafe { *t.meta.add(i) });
                    if status == FG_ST_BAD_UTF8 {
                        let _ = writeln!(stderr(), "Invalid UTF-8 input"); // line_splitter.rs:22-25, nul_splitter.rs:35-38
                        continue;
                    }
                    let res = unsafe { materialise(decoder.format(), decoder.ltsv(), &t, &buf, offs[i] as usize, i) }.and_then(|r| encoder.encode(r));
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
                buf.drain(..used as usize); // an unterminated tail waits for more bytes
                framed = false; // (the library consumes up to the last terminator: what is left holds no frame)
                debug_assert!(!has_frame(&buf, self.framing));
            }
            if last {
                if how == Fill::Idle {
                    // line_splitter.rs:26-33 / nul_splitter.rs:22-29; the partial line goes with the iterator, as in the reference
                    let _ = writeln!(stderr(), "Client hasn't sent any data for a while - Closing idle connection");
                }
                return;
            }
        }
    }

    pub fn run_transcode<T: Read>(&self, reader: BufReader<T>, tx: SyncSender<Vec<u8>>, decoder: &GpuDecoder, enc: &fg_encode_cfg) {
        let mut buf: Vec<u8> = Vec::with_capacity(2 * MAX_BYTES);
        let mut reader = widen(reader, &mut buf);
        let mut framed = has_frame(&buf, self.framing);
        // what the input's reader had buffered may already hold complete frames: they are decoded BEFORE the source is read again -- `fill`
        // blocks in `fill_buf` up to the socket's read timeout, and nothing decodable may wait behind that
        let mut carried = framed;
        loop {
            let how = if carried { Fill::Data } else { fill(&mut reader, &mut buf, MAX_BYTES, self.framing, &mut framed) };
            carried = false;
            let last = how != Fill::Data;
            let eof = how == Fill::Eof;
            if !buf.is_empty() && (eof || framed) {
                let nbytes = buf.len();
                buf.resize(nbytes + 16, 0);
                let mut r: fg_transcoded = unsafe { std::mem::zeroed() };
                let rc = unsafe {
                    fg_transcode_batch(decoder.raw_ctx(), decoder.format().raw(), self.framing, enc, buf.as_ptr(), nbytes as u64, ptr::null(), 0, eof as i32, &mut r)
                };
                assert_eq!(rc, FG_OK, "fg_transcode_batch failed: {}", rc);
                let (out, offs, meta, es, frames) = unsafe {
                    (host_slice(r.out, r.out_bytes), host_slice(r.out_offsets, r.n + 1), host_slice(r.meta, r.n), host_slice(r.enc_status, r.n), host_slice(r.frame_offsets, r.n + 1))
                };
                // only the meta column came back: fg_tables_stdout takes a failed row's count from its facility byte
                let mut only_meta: fg_tables = unsafe { std::mem::zeroed() };
                only_meta.n = r.n;
                only_meta.meta = r.meta as *mut u32;
                print_side_effects(decoder, self.framing, &buf, frames, &only_meta, r.n);
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
                buf.drain(..r.consumed as usize);
                framed = false; // (see run_decode)
                debug_assert!(!has_frame(&buf, self.framing));
            }
            if last {
                if how == Fill::Idle {
                    let _ = writeln!(stderr(), "Client hasn't sent any data for a while - Closing idle connection");
                }
                return;
            }
        }
    }
}

/// The decoder's stdout side effects for the rows of a batch (`ltsv_decoder.rs:99`: `println!("Missing value for name '{}'")`):
/// rows flagged FG_F_LTSV_NOVALUE, text rebuilt by the library from the frames.
fn print_side_effects(decoder: &GpuDecoder, framing: fg_framing, buf: &[u8], offs: &[u64], t: &fg_tables, n: u64) {
    if n == 0 || decoder.format().raw() != FG_LTSV {
        return;
    }
    let need = unsafe { fg_tables_stdout(decoder.format().raw(), framing, buf.as_ptr(), offs.as_ptr(), t, 0, n, ptr::null_mut(), 0) };
    if need <= 0 {
        return;
    }
    let mut text = vec![0u8; need as usize];
    unsafe { fg_tables_stdout(decoder.format().raw(), framing, buf.as_ptr(), offs.as_ptr(), t, 0, n, text.as_mut_ptr(), need as u64) };
    let _ = stdout().write_all(&text);
}

/// Is there at least one complete frame in `buf`?  (A batch without one would be a GPU call for nothing.)
fn has_frame(buf: &[u8], framing: fg_framing) -> bool {
    let delim = if framing == FG_FRAME_NUL { 0u8 } else { b'\n' };
    buf.contains(&delim)
}

/// Append what the source has to `buf`: blocks for the first bytes (the read timeout of the socket applies), then keeps taking
/// what `BufReader` gets per read while each read fills its whole buffer -- a SHORT read means the peer has nothing more in
/// flight, and the batch goes out.  A FULL read says nothing either way (the peer may have sent exactly `cap` bytes and gone
/// quiet), and the next `fill_buf()` would block on the socket for up to `input.timeout` with complete lines sitting undecoded:
/// so a full read only reads on while `buf` holds NO complete frame yet -- never block while there is something to decode
/// (the reference hands every line to the decoder as soon as `lines()` yields it, line_splitter.rs:17).  The batch
/// size under sustained load is therefore the reader's capacity -- READER_CAPACITY, 1 MiB = 4000 lines of 256 bytes per GPU call:
/// `widen()` sees to it whatever reader the input built.
/// `framed`: in / out, "buf holds a complete frame".  Only the bytes added here are searched for a terminator.
fn fill<T: Read>(reader: &mut BufReader<T>, buf: &mut Vec<u8>, max_bytes: usize, framing: fg_framing, framed: &mut bool) -> Fill {
    let cap = reader.capacity();
    let mut added = 0usize;
    loop {
        let chunk = match reader.fill_buf() {
            Ok(c) => c,
            Err(ref e) if e.kind() == ErrorKind::Interrupted => continue, // line_splitter.rs:21
            Err(ref e) if e.kind() == ErrorKind::WouldBlock => return Fill::Idle,
            Err(_) => return Fill::Error,
        };
        let k = chunk.len();
        if k == 0 {
            return Fill::Eof;
        }
        *framed = *framed || has_frame(chunk, framing);
        buf.extend_from_slice(chunk);
        reader.consume(k);
        added += k;
        // a short read: nothing more is pending -- or the batch is full -- or there is a complete frame to decode and the next
        // read might block
        if k < cap || added >= max_bytes || *framed {
            return Fill::Data;
        }
    }
}

#[cfg(test)]
mod tests {
    use super::*;
    use std::io::{self, Read};

    /// yields its pieces one per `read`; a read after the last piece is the peer going quiet: the test fails instead of blocking
    struct Pieces {
        pieces: Vec<Vec<u8>>,
        next: usize,
    }
    impl Read for Pieces {
        fn read(&mut self, out: &mut [u8]) -> io::Result<usize> {
            assert!(self.next < self.pieces.len(), "fill() read again while complete frames were buffered: it would have blocked");
            let p = &self.pieces[self.next];
            assert!(p.len() <= out.len());
            out[..p.len()].copy_from_slice(p);
            self.next += 1;
            Ok(p.len())
        }
    }

    #[test]
    fn a_read_that_exactly_fills_the_reader_does_not_block_on_buffered_lines() {
        let cap = 64usize;
        let mut piece = vec![b'x'; cap];
        piece[10] = b'\n';
        piece[cap - 1] = b'\n';
        let mut reader = BufReader::with_capacity(cap, Pieces { pieces: vec![piece], next: 0 });
        let mut buf = Vec::new();
        let mut framed = false;
        assert!(fill(&mut reader, &mut buf, 1 << 20, FG_FRAME_LINE, &mut framed) == Fill::Data);
        assert!(framed);
        assert_eq!(buf.len(), cap);
    }

    #[test]
    fn a_full_read_without_a_frame_reads_on() {
        let cap = 64usize;
        let first = vec![b'x'; cap]; // no terminator yet: nothing to decode, reading on is the only way forward
        let mut second = vec![b'y'; 20];
        second[19] = b'\n';
        let mut reader = BufReader::with_capacity(cap, Pieces { pieces: vec![first, second], next: 0 });
        let mut buf = Vec::new();
        let mut framed = false;
        assert!(fill(&mut reader, &mut buf, 1 << 20, FG_FRAME_LINE, &mut framed) == Fill::Data);
        assert!(framed);
        assert_eq!(buf.len(), cap + 20);
    }

    #[test]
    fn the_inputs_small_reader_is_widened_and_its_buffered_bytes_are_kept() {
        let mut piece = vec![b'z'; 40];
        piece[39] = b'\n';
        let mut small = BufReader::with_capacity(64, Pieces { pieces: vec![piece.clone(), b"tail\n".to_vec()], next: 0 });
        assert_eq!(small.fill_buf().unwrap().len(), 40); // the input's reader has read ahead
        let mut carry = Vec::new();
        let mut wide = widen(small, &mut carry);
        assert_eq!(carry, piece);
        assert!(wide.capacity() >= READER_CAPACITY);
        let mut framed = has_frame(&carry, FG_FRAME_LINE);
        assert!(framed);
        assert!(fill(&mut wide, &mut carry, 1 << 20, FG_FRAME_LINE, &mut framed) == Fill::Data);
        assert_eq!(carry.len(), 45);
    }
}
