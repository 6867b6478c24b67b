"""Host-side mirror of flowgger's Encoder / Merger interface for the GPU encode path.

Reference: ``trait Encoder { fn encode(&self, record: Record) -> Result<Vec<u8>, &'static str> }``
(src/flowgger/encoder/mod.rs:54-56) with GelfEncoder / LTSVEncoder / RFC5424Encoder / RFC3164Encoder /
PassthroughEncoder, and ``trait Merger { fn frame(&self, bytes: &mut Vec<u8>) }`` (merger/mod.rs:30-32) with
LineMerger / NulMerger / SyslenMerger.  Here one call encodes AND frames a whole decoded batch on the GPU straight
from the decode tables (fg_encode_device): the result is one contiguous byte stream in input order, which is what
the outputs write.  Configuration keys are the reference's (output.gelf_extra, output.ltsv_extra,
output.syslog_prepend_timestamp -- the latter as the already formatted header, since it is the wall clock).
There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

from . import _lib as L
from .tables import DeviceTables

MERGERS = {None: L.FG_MERGE_NONE, "none": L.FG_MERGE_NONE, "line": L.FG_MERGE_LINE, "nul": L.FG_MERGE_NUL, "syslen": L.FG_MERGE_SYSLEN}


class Encoder:
    """Base class; subclasses fix ``enc`` (fg_encoder)."""
    enc = -1
    extra_key = None  # the output.* table with extra pairs, if the encoder has one

    def __init__(self, config: Optional[dict] = None, merger: Optional[str] = None, prepend: Optional[str] = None):
        out = (config or {}).get("output", {})
        extra = out.get(self.extra_key) if self.extra_key else None
        # the reference iterates a toml Table = BTreeMap: sorted by key
        self.extra = sorted((extra or {}).items())
        for k, v in self.extra:
            if not isinstance(v, str):  # gelf_encoder.rs:27-29 / ltsv_encoder.rs:21-23
                raise TypeError(f"output.{self.extra_key} values must be strings")
        self.merger = MERGERS[merger if merger is not None else out.get("framing")]
        self.prepend = prepend

    def _cfg_struct(self, now_ts: float = 0.0):
        """fg_encode_cfg for this encoder (+ the ctypes arrays that must stay alive while it is used)."""
        ks = (C.c_char_p * max(len(self.extra), 1))(*[k.encode() for k, _ in self.extra])
        vs = (C.c_char_p * max(len(self.extra), 1))(*[v.encode() for _, v in self.extra])
        cfg = L.fg_encode_cfg(self.enc, self.merger, len(self.extra), C.cast(ks, C.POINTER(C.c_char_p)),
                              C.cast(vs, C.POINTER(C.c_char_p)), None if self.prepend is None else self.prepend.encode(), now_ts)
        return cfg, (ks, vs)

    def encode_device(self, decoder, d_bytes, d_offsets, n: int, tables: DeviceTables, now_ts: float = 0.0, stream=None,
                      want_status: bool = False, out=None):
        """Encode + frame the n decoded lines of `tables` (produced by `decoder` from d_bytes / d_offsets).
        Returns (d_out uint8, d_out_offsets int64[n+1][, d_status uint8[n]]): message i = d_out[off[i]:off[i+1]],
        empty when the line's decode or encode failed (status 1 / fg_encode_error_string).
        `out` = an optional preallocated uint8 device tensor: one call (count, scan, write) when it is large
        enough; otherwise a sizing call comes first."""
        import torch

        if stream is None:
            stream = torch.cuda.current_stream(d_bytes.device)
        cfg, _keep = self._cfg_struct(now_ts)
        d_off = torch.empty(n + 1, dtype=torch.int64, device=d_bytes.device)
        d_st = torch.empty(max(n, 1), dtype=torch.uint8, device=d_bytes.device) if want_status else None
        total = C.c_uint64()
        args = (decoder._ctx, decoder.fmt, C.byref(cfg), d_bytes.data_ptr(), d_bytes.numel(), d_offsets.data_ptr(), n,
                C.byref(tables.struct))
        tail = (d_off.data_ptr(), d_st.data_ptr() if want_status else None, C.byref(total), C.c_void_p(stream.cuda_stream))
        rc = L.FG_ERR_ENT_OVERFLOW
        if out is not None:
            rc = L.lib().fg_encode_device(*args, out.data_ptr(), out.numel(), *tail)
            if rc not in (L.FG_OK, L.FG_ERR_ENT_OVERFLOW):
                L.check(rc, "fg_encode_device")
        if rc == L.FG_ERR_ENT_OVERFLOW:
            if out is None:
                L.check(L.lib().fg_encode_device(*args, None, 0, *tail), "fg_encode_device (size)")
            out = torch.empty(max(int(total.value), 1), dtype=torch.uint8, device=d_bytes.device)
            L.check(L.lib().fg_encode_device(*args, out.data_ptr(), out.numel(), *tail), "fg_encode_device")
        res = (out[:int(total.value)], d_off)
        return res + (d_st[:n],) if want_status else res

    def encode_device_async(self, decoder, d_bytes, d_offsets, n: int, tables: DeviceTables, out, now_ts: float = 0.0, stream=None,
                            ent_hint: int = 0xFFFFFFFFFFFFFFFF):
        """fg_encode_device_async: count, scan and write queued on `stream`, no host synchronisation.  Returns (d_out_offsets
        int64[n+1], d_status uint8[n]); d_out_offsets[n] (device) = the bytes the batch needs -- when that exceeds out.numel()
        nothing was written to `out`.  ent_hint: an upper bound of the entries in `tables` (0 = none)."""
        import torch

        if stream is None:
            stream = torch.cuda.current_stream(d_bytes.device)
        cfg, _keep = self._cfg_struct(now_ts)
        d_off = torch.empty(n + 1, dtype=torch.int64, device=d_bytes.device)
        d_st = torch.empty(max(n, 1), dtype=torch.uint8, device=d_bytes.device)
        L.check(L.lib().fg_encode_device_async(decoder._ctx, decoder.fmt, C.byref(cfg), d_bytes.data_ptr(), d_bytes.numel(), d_offsets.data_ptr(),
                                               n, C.byref(tables.struct), out.data_ptr(), out.numel(), d_off.data_ptr(), d_st.data_ptr(),
                                               ent_hint, C.c_void_p(stream.cuda_stream)), "fg_encode_device_async")
        return d_off, d_st[:n]

    @staticmethod
    def error_string(status: int) -> Optional[str]:
        s = L.lib().fg_encode_error_string(status)
        return None if s is None else s.decode()


class GelfEncoder(Encoder):  # encoder/gelf_encoder.rs
    enc, extra_key = L.FG_ENC_GELF, "gelf_extra"


class LTSVEncoder(Encoder):  # encoder/ltsv_encoder.rs
    enc, extra_key = L.FG_ENC_LTSV, "ltsv_extra"


class RFC5424Encoder(Encoder):  # encoder/rfc5424_encoder.rs
    enc = L.FG_ENC_RFC5424


class RFC3164Encoder(Encoder):  # encoder/rfc3164_encoder.rs
    enc = L.FG_ENC_RFC3164


class PassthroughEncoder(Encoder):  # encoder/passthrough_encoder.rs
    enc = L.FG_ENC_PASSTHROUGH


class Transcoded:
    """Result of Pipeline.run: the encoded + framed stream and one verdict per line (copies of ctx-owned pinned memory)."""

    def __init__(self, out, out_offsets, meta, enc_status, frame_offsets, consumed):
        self.out, self.out_offsets, self.meta, self.enc_status = out, out_offsets, meta, enc_status
        self.frame_offsets, self.consumed = frame_offsets, consumed

    @property
    def n(self) -> int:
        return len(self.enc_status)

    @property
    def dec_status(self):
        return (self.meta & 0xFF).astype("uint8")

    def message(self, i: int) -> bytes:
        return self.out[int(self.out_offsets[i]):int(self.out_offsets[i + 1])].tobytes()


class Pipeline:
    """decoder -> encoder -> merger for whole batches with HOST buffers (fg_transcode_batch): what
    ``handle_line`` (splitter/line_splitter.rs:44-54) does per line.  Only the encoded bytes and the per-line
    verdicts come back over PCIe; the decode tables stay in HBM."""

    def __init__(self, decoder, encoder: Encoder):
        self.decoder, self.encoder = decoder, encoder

    def _call(self, framing: int, data, nbytes: int, offsets, n: int, final: bool, now_ts: float) -> Transcoded:
        import numpy as np

        cfg, _keep = self.encoder._cfg_struct(now_ts)
        res = L.fg_transcoded()
        rc = L.lib().fg_transcode_batch(self.decoder._ctx, self.decoder.fmt, framing, C.byref(cfg), data.ctypes.data, nbytes,
                                        None if offsets is None else offsets.ctypes.data, n, int(final), C.byref(res))
        L.check(rc, "fg_transcode_batch")
        m = int(res.n)

        def view(ptr, count, dt):
            if not count or not ptr:
                return np.zeros(0, dt)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), (count * np.dtype(dt).itemsize,)).view(dt).copy()

        return Transcoded(view(res.out, int(res.out_bytes), np.uint8),
                          view(res.out_offsets, m + 1 if m else 0, np.uint64) if m else np.zeros(1, np.uint64),
                          view(res.meta, m, np.uint32), view(res.enc_status, m, np.uint8),
                          view(res.frame_offsets, m + 1, np.uint64) if (m and res.frame_offsets) else None, int(res.consumed))

    def run_packed(self, data, offsets, now_ts: float = 0.0) -> Transcoded:
        """framed lines (packed bytes + offsets[n + 1], as produced by pack_lines)"""
        import numpy as np

        data = np.ascontiguousarray(data, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        return self._call(L.FG_FRAME_NONE, data, int(offsets[-1]) if n else 0, offsets, n, True, now_ts)

    def run_stream(self, raw, framing: int, final: bool = True, now_ts: float = 0.0) -> Transcoded:
        """a raw stream chunk, framed on the GPU ("\\n" / NUL); bytes past `.consumed` belong to the next chunk"""
        import numpy as np

        data = np.frombuffer(raw, dtype=np.uint8) if isinstance(raw, (bytes, bytearray)) else np.ascontiguousarray(raw, dtype=np.uint8)
        if len(data) == 0:
            data = np.zeros(1, np.uint8)
            return self._call(framing, data, 0, None, 0, final, now_ts)
        return self._call(framing, data, len(data), None, 0, final, now_ts)
