"""Decoders -- host-side mirror of flowgger's `Decoder` trait for the GPU bulk-parse stage.

Reference interface (src/flowgger/decoder/mod.rs:23-46):

    pub trait Decoder: CloneBoxedDecoder {
        fn decode(&self, line: &str) -> Result<Record, &'static str>;
    }

Same names, same argument meaning, same error strings: ``decode(line)`` returns a
:class:`~flowgger_amd.record.Record` or raises :class:`~flowgger_amd.record.DecodeError` whose
message is the reference's ``&'static str``.  What is new is ``decode_batch`` / ``decode_packed``:
the batching framer hands over N framed lines at once and ONE call runs the gfx950 kernels
(the per-line loop it replaces: src/flowgger/splitter/line_splitter.rs:17,44-54).

Every call goes through the C ABI of libfg_hip.so; there is no CPU implementation here.
"""
from __future__ import annotations

import ctypes as C
from typing import Iterable, List, Optional, Sequence, Tuple, Union

import numpy as np

from . import _lib as L
from .record import DecodeError, Record, parse_canonical
from .tables import DeviceTables, HostTables

_TYPE_IDS = {"string": L.FG_T_STRING, "bool": L.FG_T_BOOL, "f64": L.FG_T_F64, "i64": L.FG_T_I64, "u64": L.FG_T_U64}


def pack_lines(lines: Iterable[Union[str, bytes]]) -> Tuple[np.ndarray, np.ndarray]:
    """The batching framer's container: framed lines -> (packed bytes, offsets[n+1] uint64)."""
    bl = [ln.encode("utf-8", "surrogateescape") if isinstance(ln, str) else bytes(ln) for ln in lines]
    offsets = np.zeros(len(bl) + 1, np.uint64)
    if bl:
        offsets[1:] = np.cumsum([len(b) for b in bl], dtype=np.uint64)
    data = np.frombuffer(b"".join(bl), np.uint8).copy() if bl else np.zeros(0, np.uint8)
    return data, offsets


class Decoder:
    """Base: owns an fg_ctx (the analogue of Box<dyn Decoder + Send>)."""

    fmt: int = -1

    def __init__(self, config: Optional[dict] = None, device: int = 0):
        self._cfg = self._make_cfg(config)
        self._ctx = C.c_void_p()
        self.device = device
        L.check(L.lib().fg_create(device, C.byref(self._cfg) if self._cfg is not None else None,
                                  C.byref(self._ctx)), "fg_create")

    # -- configuration hook (only LTSV has one) ------------------------------------------
    def _make_cfg(self, config: Optional[dict]):
        return None

    def clone_boxed(self) -> "Decoder":  # decoder/mod.rs:29-36
        other = object.__new__(type(self))
        other._cfg = self._cfg
        other._keep = getattr(self, "_keep", None)
        other.device = self.device
        other._ctx = C.c_void_p()
        L.check(L.lib().fg_clone(self._ctx, C.byref(other._ctx)), "fg_clone")
        return other

    def set_launch_opts(self, lines_per_group: int = 0, tile_cap: int = 0, waves_per_cu: int = 0, gelf_lds_budget: int = 0,
                        gelf_window_kib: int = 0, gelf_generic: bool = False, transcode_one_piece: bool = False, chunk_lines: int = 0,
                        no_head: bool = False, force_head: bool = False, sd_walk: bool = False, sd_pairs: bool = False, no_zero_copy: bool = False, frame_kernel_upload: bool = False, frame_classic: bool = False, static_chunks: bool = False, frame_selftest_stall: bool = False, ent_chunk: int = 0, no_taper: bool = False, taper_levels: int = 0, no_fused_framing: bool = False, fused_look: int = 0, fused_ext: int = 0, rfc3164_regroup: int = 0) -> None:
        """fg_set_launch_opts: launch-geometry overrides of this ctx (parity sweeps over the kernel variants, tuning); 0 = the
        library's own choice.  Results are identical for every setting.  (The library reads no environment variables.)"""
        lo = L.fg_launch_opts(lines_per_group, tile_cap, waves_per_cu, gelf_lds_budget, gelf_window_kib,
                              (L.FG_LO_GELF_GENERIC if gelf_generic else 0) | (L.FG_LO_TRANSCODE_ONE_PIECE if transcode_one_piece else 0) |
                              (L.FG_LO_NO_HEAD if no_head else 0) | (L.FG_LO_FORCE_HEAD if force_head else 0) |
                              (L.FG_LO_SD_WALK if sd_walk else 0) | (L.FG_LO_SD_PAIRS if sd_pairs else 0) | (L.FG_LO_NO_ZERO_COPY if no_zero_copy else 0) | (L.FG_LO_FRAME_KERNEL_UPLOAD if frame_kernel_upload else 0) | (L.FG_LO_FRAME_CLASSIC if frame_classic else 0) | (L.FG_LO_STATIC_CHUNKS if static_chunks else 0) | (L.FG_LO_FRAME_SELFTEST_STALL if frame_selftest_stall else 0) | (L.FG_LO_NO_TAPER if no_taper else 0) | (L.FG_LO_TAPER_1 if taper_levels & 1 else 0) | (L.FG_LO_TAPER_2 if taper_levels & 2 else 0) | (L.FG_LO_NO_FUSED_FRAMING if no_fused_framing else 0) | (L.FG_LO_RFC3164_REGROUP if rfc3164_regroup == 1 else L.FG_LO_RFC3164_NO_REGROUP if rfc3164_regroup == 2 else 0),
                              chunk_lines, ent_chunk, fused_look, fused_ext)
        L.check(L.lib().fg_set_launch_opts(self._ctx, C.byref(lo)), "fg_set_launch_opts")

    def close(self) -> None:
        if getattr(self, "_ctx", None) is not None and self._ctx.value:
            L.lib().fg_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    # -- the reference interface ------------------------------------------------------------
    def decode(self, line: Union[str, bytes]) -> Record:
        res = self.decode_batch([line])[0]
        if isinstance(res, DecodeError):
            raise res
        return res

    # -- bulk interface -----------------------------------------------------------------------
    def decode_packed(self, data: np.ndarray, offsets: np.ndarray) -> HostTables:
        """Host buffers in, host tables out (H2D + kernels + D2H inside the library)."""
        data = np.ascontiguousarray(data, np.uint8)
        offsets = np.ascontiguousarray(offsets, np.uint64)
        n = len(offsets) - 1
        st = L.fg_tables()
        L.check(L.lib().fg_decode_batch(self._ctx, self.fmt, data.ctypes.data, data.size,
                                        offsets.ctypes.data, n, C.byref(st)), "fg_decode_batch")
        return HostTables.from_struct(st)

    def decode_batch(self, lines: Sequence[Union[str, bytes]]) -> List[Union[Record, DecodeError]]:
        data, offsets = pack_lines(lines)
        tab = self.decode_packed(data, offsets)
        n_rows = len(offsets) - 1
        if self.fmt == L.FG_LTSV:  # the decoder's stdout side effect (ltsv_decoder.rs:99), reproduced from the rows' flags
            import sys

            from .tables import tables_stdout

            # (only when a row carries the flag; fg_tables_stdout reads [offsets[i], offsets[i+1]) only: no copy, no slack -- ADVICE r3)
            if n_rows and bool(((tab.a["meta"][:n_rows] >> 24) & L.FG_F_LTSV_NOVALUE).any()):
                text = tables_stdout(tab, self.fmt, data, offsets)
                if text:
                    sys.stdout.write(text.decode("utf-8", "replace"))
        blob, offs = tab.serialize(self.fmt, data, offsets, cfg=self._cfg)
        raw = blob.tobytes()
        return [parse_canonical(raw[int(offs[i]):int(offs[i + 1])]) for i in range(len(lines))]

    def decode_device(self, d_bytes, d_offsets, tables: DeviceTables, stream=None) -> None:
        """Device-resident hot path: torch uint8 / int64|uint64 tensors already in HBM.
        Asynchronous on `stream` (a torch.cuda.Stream; default = torch's current stream)."""
        import torch

        if stream is None:
            stream = torch.cuda.current_stream(d_bytes.device)
        n = d_offsets.numel() - 1
        L.check(L.lib().fg_decode_batch_device(self._ctx, self.fmt, d_bytes.data_ptr(), d_bytes.numel(),
                                               d_offsets.data_ptr(), n, C.byref(tables.struct),
                                               C.c_void_p(stream.cuda_stream)), "fg_decode_batch_device")

    # -- GPU framing (the splitter side) -------------------------------------------
    def frame_device(self, d_bytes, framing: int, cap_frames: Optional[int] = None, stream=None):
        """Frame a raw byte stream resident in HBM: `buf_reader.lines()` (framing = FG_FRAME_LINE) or
        `buf_reader.split(0)` (FG_FRAME_NUL) plus the per-frame `str::from_utf8` check
        (splitter/line_splitter.rs:17-25, nul_splitter.rs:18-40).  Returns (d_offsets, d_bad_utf8, n):
        frame i = d_bytes[offsets[i]:offsets[i+1]] INCLUDING its terminator."""
        import torch

        if stream is None:
            stream = torch.cuda.current_stream(d_bytes.device)
        nbytes = d_bytes.numel()
        cap = cap_frames if cap_frames is not None else nbytes // 16 + 16
        while True:
            d_offsets = torch.empty(cap + 2, dtype=torch.int64, device=d_bytes.device)
            d_bad = torch.empty(cap + 1, dtype=torch.uint8, device=d_bytes.device)
            n = C.c_uint64()
            rc = L.lib().fg_frame_device(self._ctx, framing, d_bytes.data_ptr(), nbytes, d_offsets.data_ptr(),
                                         d_bad.data_ptr(), cap, C.byref(n), C.c_void_p(stream.cuda_stream))
            if rc == L.FG_ERR_ENT_OVERFLOW:
                cap = int(n.value) + 16
                continue
            L.check(rc, "fg_frame_device")
            return d_offsets, d_bad, int(n.value)

    def decode_frames_device(self, d_bytes, d_offsets, n: int, tables: DeviceTables, framing: int, d_bad=None,
                             stream=None) -> None:
        """Decode frames produced by frame_device (terminators are stripped inside the kernels; frames
        flagged in d_bad get status FG_ST_BAD_UTF8 = the reference's "Invalid UTF-8 input")."""
        import torch

        if stream is None:
            stream = torch.cuda.current_stream(d_bytes.device)
        L.check(L.lib().fg_decode_frames_device(self._ctx, self.fmt, framing, d_bytes.data_ptr(), d_bytes.numel(),
                                                d_offsets.data_ptr(), n, d_bad.data_ptr() if d_bad is not None else None,
                                                C.byref(tables.struct), C.c_void_p(stream.cuda_stream)),
                "fg_decode_frames_device")

    def frame_decode_device(self, d_bytes, framing: int, tables: DeviceTables, cap_frames: int, final: bool = True, avg_line: int = 0,
                            stream=None):
        """Frame AND decode a raw byte stream resident in HBM in ONE kernel (fg_frame_decode_device: the decode kernels frame their
        tiles themselves; LineSplitter::run / NulSplitter::run for a chunk of the stream).  Asynchronous.  Returns
        (d_offsets int64[cap_frames + 2], d_result int64[2]): d_result[0] = frames found (rows beyond cap_frames were not written),
        d_result[1] != 0 = the launch gave up (use frame_device + decode_frames_device)."""
        import torch

        if stream is None:
            stream = torch.cuda.current_stream(d_bytes.device)
        d_offsets = torch.empty(cap_frames + 2, dtype=torch.int64, device=d_bytes.device)
        d_result = torch.zeros(2, dtype=torch.int64, device=d_bytes.device)
        L.check(L.lib().fg_frame_decode_device(self._ctx, self.fmt, framing, d_bytes.data_ptr(), d_bytes.numel(), int(final),
                                               d_offsets.data_ptr(), cap_frames, C.byref(tables.struct), avg_line, d_result.data_ptr(),
                                               C.c_void_p(stream.cuda_stream)), "fg_frame_decode_device")
        return d_offsets, d_result

    def frame_decode_batch(self, raw: Union[bytes, np.ndarray], framing: int, final: bool = True):
        """Raw chunk of the byte stream in, tables out: GPU framing + UTF-8 validation + decode in one
        call (fg_frame_decode_batch; the body of LineSplitter::run / NulSplitter::run).  Returns
        (HostTables, offsets uint64[n+1], consumed): frame i = raw[offsets[i]:offsets[i+1]] including
        its terminator; raw[consumed:] is an unterminated tail to carry over (empty when `final`)."""
        buf = np.frombuffer(raw, np.uint8) if not isinstance(raw, np.ndarray) else np.ascontiguousarray(raw, np.uint8)
        padded = np.zeros(buf.size + 16, np.uint8)
        padded[:buf.size] = buf
        st = L.fg_tables()
        off = C.c_void_p()
        n, used = C.c_uint64(), C.c_uint64()
        L.check(L.lib().fg_frame_decode_batch(self._ctx, self.fmt, framing, padded.ctypes.data, buf.size, int(final),
                                              C.byref(st), C.byref(off), C.byref(n), C.byref(used)), "fg_frame_decode_batch")
        nf = int(n.value)
        if nf == 0:
            return None, np.zeros(1, np.uint64), int(used.value)
        offsets = np.ctypeslib.as_array(C.cast(off, C.POINTER(C.c_uint64)), (nf + 1,)).copy()
        return HostTables.from_struct(st), offsets, int(used.value)

    # -- GELF encoder from the tables ------------------------------------------------
    def encode_gelf_device(self, d_bytes, d_offsets, n: int, tables: DeviceTables, extra: Optional[dict] = None, stream=None):
        """GelfEncoder::encode (encoder/gelf_encoder.rs:59-115) for every decoded line of `tables`, on the
        GPU, without materialising Records.  `extra` = the output.gelf_extra table.  Returns
        (d_out uint8, d_out_offsets int64[n+1]): JSON of line i = d_out[off[i]:off[i+1]] (empty for a
        line whose decode failed)."""
        import torch

        if stream is None:
            stream = torch.cuda.current_stream(d_bytes.device)
        items = list((extra or {}).items())
        ks = (C.c_char_p * max(len(items), 1))(*[k.encode() for k, _ in items])
        vs = (C.c_char_p * max(len(items), 1))(*[v.encode() for _, v in items])
        ex = L.fg_gelf_extra(len(items), C.cast(ks, C.POINTER(C.c_char_p)), C.cast(vs, C.POINTER(C.c_char_p)))
        d_off = torch.empty(n + 1, dtype=torch.int64, device=d_bytes.device)
        total = C.c_uint64()
        args = (self._ctx, self.fmt, d_bytes.data_ptr(), d_bytes.numel(), d_offsets.data_ptr(), n, C.byref(tables.struct), C.byref(ex))
        L.check(L.lib().fg_encode_gelf_device(*args, None, 0, d_off.data_ptr(), C.byref(total), C.c_void_p(stream.cuda_stream)),
                "fg_encode_gelf_device (size)")
        d_out = torch.empty(max(int(total.value), 1), dtype=torch.uint8, device=d_bytes.device)
        L.check(L.lib().fg_encode_gelf_device(*args, d_out.data_ptr(), d_out.numel(), d_off.data_ptr(), C.byref(total),
                                              C.c_void_p(stream.cuda_stream)), "fg_encode_gelf_device")
        return d_out[:int(total.value)], d_off

    def set_timing(self, enabled: bool = True) -> None:
        L.check(L.lib().fg_set_timing(self._ctx, int(enabled)), "fg_set_timing")

    def last_kernel_ms(self) -> float:
        ms = C.c_float()
        L.check(L.lib().fg_last_kernel_ms(self._ctx, C.byref(ms)), "fg_last_kernel_ms")
        return float(ms.value)

    def error_string(self, status: int) -> Optional[str]:
        s = L.lib().fg_error_string(self.fmt, status)
        return None if s is None else s.decode()


class RFC5424Decoder(Decoder):
    """src/flowgger/decoder/rfc5424_decoder.rs:8-50 (stateless; config ignored, :12-14)."""
    fmt = L.FG_RFC5424


class GelfDecoder(Decoder):
    """src/flowgger/decoder/gelf_decoder.rs:10-125 (config ignored, :16-18)."""
    fmt = L.FG_GELF


class RFC3164Decoder(Decoder):
    """src/flowgger/decoder/rfc3164_decoder.rs:10-213.  The reference's config is ignored (:13-15); two things it takes
    from its environment are explicit here: ``current_year`` (it reads the clock per parse, :179; default
    FG_YEAR_NOW: the library re-reads the UTC year at every decode call, so a long-lived decoder and its clones cross
    New Year like the reference; a fixed year is for tests / replays) and the IANA zone table behind ``time_tz::timezones::get_by_name`` (:195; default: every zone
    of the system tz database, flowgger_amd/tzdb.py).  {"rfc3164": {"current_year": 2020, "zones": [...] | None}}."""
    fmt = L.FG_RFC3164

    def __init__(self, config: Optional[dict] = None, device: int = 0):
        super().__init__(config, device)
        from . import tzdb

        opt = (config or {}).get("rfc3164", {})
        self.current_year = int(opt.get("current_year", L.FG_YEAR_NOW))
        zones = opt.get("zones", "all")
        self.tz_table = tzdb.default_table() if zones == "all" else (tzdb.build_table(zones) if zones else None)
        self._apply()

    def _apply(self):
        cfg = L.fg_rfc3164_cfg(self.current_year, None)
        t = self.tz_table
        if t is not None and len(t.names):
            names = (C.c_char_p * len(t.names))(*[n.encode() for n in t.names])
            zf = np.ascontiguousarray(t.zone_first, np.uint32)
            us = np.ascontiguousarray(t.utc_start, np.int64)
            uo = np.ascontiguousarray(t.utc_offset, np.int32)
            tab = L.fg_tz_table(len(t.names), C.cast(names, C.POINTER(C.c_char_p)), zf.ctypes.data, us.ctypes.data, uo.ctypes.data)
            cfg.tz = C.pointer(tab)
            self._keep3164 = (names, zf, us, uo, tab)
        L.check(L.lib().fg_set_rfc3164(self._ctx, C.byref(cfg)), "fg_set_rfc3164")

    def clone_boxed(self) -> "Decoder":
        other = super().clone_boxed()
        other.current_year, other.tz_table = self.current_year, self.tz_table
        return other


class LTSVDecoder(Decoder):
    """src/flowgger/decoder/ltsv_decoder.rs:17-221.  `config` mirrors the TOML tables
    ``input.ltsv_schema`` (name -> "string|bool|f64|i64|u64", case-insensitive, :33-45) and
    ``input.ltsv_suffixes`` (type -> suffix, :57-81): {"input": {"ltsv_schema": {...},
    "ltsv_suffixes": {...}}}."""
    fmt = L.FG_LTSV

    def _make_cfg(self, config: Optional[dict]):
        inp = (config or {}).get("input", {})
        schema = inp.get("ltsv_schema")
        suffixes = inp.get("ltsv_suffixes")
        cfg = L.fg_cfg()
        keep = []
        names, types = [], []
        if schema is not None:
            if not isinstance(schema, dict):
                raise ValueError("input.ltsv_schema must be a list of key/type pairs")
            for name, sdtype in schema.items():
                if not isinstance(sdtype, str):
                    raise ValueError("input.ltsv_schema types must be strings")
                t = _TYPE_IDS.get(sdtype.lower())
                if t is None:
                    raise ValueError(f"Unsupported type in input.ltsv_schema for name [{name}]")
                names.append(name.encode())
                types.append(t)
        cfg.n_schema = len(names)
        arr_n = (C.c_char_p * max(len(names), 1))(*names)
        arr_t = (C.c_uint8 * max(len(types), 1))(*types)
        cfg.schema_names = C.cast(arr_n, C.POINTER(C.c_char_p))
        cfg.schema_types = C.cast(arr_t, C.POINTER(C.c_uint8))
        keep += [arr_n, arr_t, names]
        if suffixes is not None:
            if not isinstance(suffixes, dict):
                raise ValueError("input.ltsv_suffixes must be a list of type/suffixes pairs")
            for sdtype, suffix in suffixes.items():
                if not isinstance(suffix, str):
                    raise ValueError("input.ltsv_suffixes suffixes must be strings")
                t = sdtype.lower()
                if t == "string":
                    raise ValueError("Strings cannot be suffixed")
                if t not in ("bool", "f64", "i64", "u64"):
                    raise ValueError(f"Unsupported type in input.ltsv_suffixes for type [{sdtype}]")
                setattr(cfg, "suffix_" + t, suffix.encode())
        self._keep = keep
        return cfg
