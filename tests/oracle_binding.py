"""ctypes binding of oracle/libfg_oracle.so -- the CPU oracle (test infrastructure only)."""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
ORACLE_DIR = ROOT / "oracle"
LIB = ORACLE_DIR / "libfg_oracle.so"
RFC5424, LTSV, GELF, RFC3164 = 0, 1, 2, 3
_TYPE_IDS = {"string": 0, "bool": 1, "f64": 2, "i64": 3, "u64": 4}


ENC_GELF, ENC_LTSV, ENC_RFC5424, ENC_RFC3164, ENC_PASSTHROUGH = 0, 1, 2, 3, 4
MERGE_NONE, MERGE_LINE, MERGE_NUL, MERGE_SYSLEN = 0, 1, 2, 3


class fgo_enc_opts(C.Structure):
    _fields_ = [("extra_keys", C.POINTER(C.c_char_p)), ("extra_vals", C.POINTER(C.c_char_p)), ("n_extra", C.c_uint32),
                ("prepend", C.c_char_p), ("now_ts", C.c_double)]


class fgo_ltsv_cfg(C.Structure):
    _fields_ = [("schema_names", C.POINTER(C.c_char_p)), ("schema_types", C.POINTER(C.c_uint8)),
                ("n_schema", C.c_uint32), ("suffix_bool", C.c_char_p), ("suffix_f64", C.c_char_p),
                ("suffix_i64", C.c_char_p), ("suffix_u64", C.c_char_p)]


def build_oracle() -> Path:
    src = ORACLE_DIR / "fg_oracle.cpp"
    if not LIB.exists() or LIB.stat().st_mtime < max(src.stat().st_mtime, (ORACLE_DIR / "fg_oracle.h").stat().st_mtime):
        subprocess.run(["make", "-C", str(ORACLE_DIR)], check=True, capture_output=True)
    return LIB


class Oracle:
    def __init__(self):
        self.lib = C.CDLL(str(build_oracle()))
        L = self.lib
        vp, u64 = C.c_void_p, C.c_uint64
        L.fgo_decode.restype = C.c_int64
        L.fgo_decode.argtypes = [C.c_int, vp, vp, u64, vp, u64]
        L.fgo_decode_batch.restype = C.c_int64
        L.fgo_decode_batch.argtypes = [C.c_int, vp, vp, vp, u64, vp, u64, vp, C.c_int]
        L.fgo_bench_decode.restype = C.c_double
        L.fgo_bench_decode.argtypes = [C.c_int, vp, vp, vp, u64, C.c_int, vp, vp]
        for f in (L.fgo_rfc3339_to_unix, L.fgo_rust_parse_f64, L.fgo_english_time_to_unix):
            f.argtypes = [vp, u64, C.POINTER(C.c_double)]
        L.fgo_json_number.argtypes = [vp, u64, C.POINTER(C.c_int), C.POINTER(u64)]
        L.fgo_gelf_encode.restype = C.c_int64
        L.fgo_gelf_encode.argtypes = [vp, u64, vp, vp, C.c_uint32, vp, u64]
        L.fgo_decode_encode_gelf_batch.restype = C.c_int64
        L.fgo_decode_encode_gelf_batch.argtypes = [C.c_int, vp, vp, vp, u64, vp, vp, C.c_uint32, vp, u64, vp]
        L.fgo_dtoa.argtypes = [C.c_double, vp, C.c_int]
        L.fgo_set_rfc3164.restype = None
        L.fgo_set_rfc3164.argtypes = [C.c_int, C.c_uint32, vp, vp, vp, vp]
        L.fgo_rust_display_f64.argtypes = [C.c_double, vp, C.c_int]
        L.fgo_encode.restype = C.c_int64
        L.fgo_encode.argtypes = [C.c_int, C.c_int, vp, u64, vp, vp, u64, C.POINTER(C.c_char_p)]
        L.fgo_decode_encode_batch.restype = C.c_int64
        L.fgo_decode_encode_batch.argtypes = [C.c_int, vp, C.c_int, C.c_int, vp, vp, u64, vp, vp, u64, vp, vp]

    def set_rfc3164(self, current_year: int, table=None):
        """table: flowgger_amd.tzdb.TzTable (None = no zone names)"""
        if table is None:
            self.lib.fgo_set_rfc3164(current_year, 0, None, None, None, None)
            return
        names = (C.c_char_p * len(table.names))(*[n.encode() for n in table.names])
        zf = np.ascontiguousarray(table.zone_first, np.uint32)
        us = np.ascontiguousarray(table.utc_start, np.int64)
        uo = np.ascontiguousarray(table.utc_offset, np.int32)
        self.lib.fgo_set_rfc3164(current_year, len(table.names), names, zf.ctypes.data, us.ctypes.data, uo.ctypes.data)

    @staticmethod
    def make_cfg(config):
        """config: same dict shape as flowgger_amd.LTSVDecoder; returns (struct|None, keepalive)."""
        if not config:
            return None, None
        inp = config.get("input", {})
        schema = inp.get("ltsv_schema") or {}
        suffixes = inp.get("ltsv_suffixes") or {}
        names = [k.encode() for k in schema]
        types = [_TYPE_IDS[v.lower()] for v in schema.values()]
        cfg = fgo_ltsv_cfg()
        an = (C.c_char_p * max(len(names), 1))(*names)
        at = (C.c_uint8 * max(len(types), 1))(*types)
        cfg.schema_names = C.cast(an, C.POINTER(C.c_char_p))
        cfg.schema_types = C.cast(at, C.POINTER(C.c_uint8))
        cfg.n_schema = len(names)
        for k, v in suffixes.items():
            setattr(cfg, "suffix_" + k.lower(), v.encode())
        return cfg, (an, at, names)

    def decode(self, fmt: int, line, config=None) -> bytes:
        b = line.encode("utf-8", "surrogateescape") if isinstance(line, str) else bytes(line)
        cfg, keep = self.make_cfg(config)
        cfgp = C.byref(cfg) if cfg is not None else None
        buf = C.create_string_buffer(len(b) * 3 + 4096)
        n = self.lib.fgo_decode(fmt, cfgp, b, len(b), buf, len(buf))
        assert 0 <= n <= len(buf)
        return buf.raw[:n]

    def decode_batch(self, fmt: int, data: np.ndarray, offsets: np.ndarray, config=None, threads: int = 8):
        data = np.ascontiguousarray(data, np.uint8)
        offsets = np.ascontiguousarray(offsets, np.uint64)
        n = len(offsets) - 1
        cfg, keep = self.make_cfg(config)
        cfgp = C.byref(cfg) if cfg is not None else None
        offs = np.zeros(n + 1, np.uint64)
        total = self.lib.fgo_decode_batch(fmt, cfgp, data.ctypes.data, offsets.ctypes.data, n, None, 0,
                                          offs.ctypes.data, threads)
        blob = np.zeros(max(total, 1), np.uint8)
        self.lib.fgo_decode_batch(fmt, cfgp, data.ctypes.data, offsets.ctypes.data, n, blob.ctypes.data, total,
                                  offs.ctypes.data, threads)
        return blob[:total], offs

    @staticmethod
    def _extra(extra):
        extra = list((extra or {}).items())
        ks = (C.c_char_p * max(len(extra), 1))(*[k.encode() for k, _ in extra])
        vs = (C.c_char_p * max(len(extra), 1))(*[v.encode() for _, v in extra])
        return ks, vs, len(extra)

    def gelf_encode(self, canonical: bytes, extra=None) -> bytes:
        """GelfEncoder::encode on a canonical Ok record; extra = the output.gelf_extra table (dict)."""
        ks, vs, n = self._extra(extra)
        need = self.lib.fgo_gelf_encode(canonical, len(canonical), ks, vs, n, None, 0)
        assert need >= 0, "not an Ok record"
        buf = C.create_string_buffer(int(need) + 1)
        self.lib.fgo_gelf_encode(canonical, len(canonical), ks, vs, n, buf, need)
        return buf.raw[:need]

    def decode_encode_gelf_batch(self, fmt: int, data: np.ndarray, offsets: np.ndarray, config=None, extra=None):
        data = np.ascontiguousarray(data, np.uint8)
        offsets = np.ascontiguousarray(offsets, np.uint64)
        n = len(offsets) - 1
        cfg, keep = self.make_cfg(config)
        cfgp = C.byref(cfg) if cfg is not None else None
        ks, vs, ne = self._extra(extra)
        offs = np.zeros(n + 1, np.uint64)
        total = self.lib.fgo_decode_encode_gelf_batch(fmt, cfgp, data.ctypes.data, offsets.ctypes.data, n, ks, vs, ne, None, 0,
                                                      offs.ctypes.data)
        blob = np.zeros(max(int(total), 1), np.uint8)
        self.lib.fgo_decode_encode_gelf_batch(fmt, cfgp, data.ctypes.data, offsets.ctypes.data, n, ks, vs, ne, blob.ctypes.data,
                                              total, offs.ctypes.data)
        return blob[:int(total)], offs

    def _opts(self, extra=None, prepend=None, now_ts=0.0):
        """extra: dict, passed in sorted key order (the configuration table is a BTreeMap)."""
        items = sorted((extra or {}).items())
        ks = (C.c_char_p * max(len(items), 1))(*[k.encode() for k, _ in items])
        vs = (C.c_char_p * max(len(items), 1))(*[v.encode() for _, v in items])
        o = fgo_enc_opts()
        o.extra_keys = C.cast(ks, C.POINTER(C.c_char_p))
        o.extra_vals = C.cast(vs, C.POINTER(C.c_char_p))
        o.n_extra = len(items)
        o.prepend = None if prepend is None else (prepend.encode() if isinstance(prepend, str) else prepend)
        o.now_ts = now_ts
        return o, (ks, vs)

    def encode(self, enc: int, canonical: bytes, merger: int = 0, extra=None, prepend=None, now_ts=0.0):
        """Encoder::encode (+ Merger::frame) on a canonical Ok record -> bytes, or the Err string (str)."""
        o, keep = self._opts(extra, prepend, now_ts)
        err = C.c_char_p()
        need = self.lib.fgo_encode(enc, merger, canonical, len(canonical), C.byref(o), None, 0, C.byref(err))
        if need == -2:
            return err.value.decode()
        assert need >= 0, "not an Ok record"
        buf = C.create_string_buffer(int(need) + 1)
        self.lib.fgo_encode(enc, merger, canonical, len(canonical), C.byref(o), buf, need, C.byref(err))
        return buf.raw[:need]

    def decode_encode_batch(self, fmt: int, enc: int, merger: int, data: np.ndarray, offsets: np.ndarray, config=None,
                            extra=None, prepend=None, now_ts=0.0):
        """-> (blob, out_offsets, status) ; status 0 Ok / 1 decode failed / 2 encode failed"""
        data = np.ascontiguousarray(data, np.uint8)
        offsets = np.ascontiguousarray(offsets, np.uint64)
        n = len(offsets) - 1
        cfg, keep = self.make_cfg(config)
        cfgp = C.byref(cfg) if cfg is not None else None
        o, keep2 = self._opts(extra, prepend, now_ts)
        offs = np.zeros(n + 1, np.uint64)
        st = np.zeros(max(n, 1), np.uint8)
        total = self.lib.fgo_decode_encode_batch(fmt, cfgp, enc, merger, data.ctypes.data, offsets.ctypes.data, n, C.byref(o),
                                                 None, 0, offs.ctypes.data, st.ctypes.data)
        blob = np.zeros(max(int(total), 1), np.uint8)
        self.lib.fgo_decode_encode_batch(fmt, cfgp, enc, merger, data.ctypes.data, offsets.ctypes.data, n, C.byref(o),
                                         blob.ctypes.data, total, offs.ctypes.data, st.ctypes.data)
        return blob[:int(total)], offs, st[:n]

    def rust_display(self, v: float) -> str:
        buf = C.create_string_buffer(512)
        n = self.lib.fgo_rust_display_f64(v, buf, 512)
        return buf.raw[:n].decode()

    def dtoa(self, v: float) -> str:
        buf = C.create_string_buffer(64)
        n = self.lib.fgo_dtoa(v, buf, 64)
        return buf.raw[:n].decode()

    def frame(self, raw: bytes, framing: str):
        """fgo_frame: the splitters' framing + UTF-8 check -> list of (start, end_with_terminator, body bytes, is_valid_utf8)"""
        data = np.frombuffer(raw, np.uint8) if raw else np.zeros(1, np.uint8)
        self.lib.fgo_frame.restype = C.c_int64
        fr = 1 if framing == "line" else 2
        vp = C.c_void_p
        n = self.lib.fgo_frame(C.c_int(fr), vp(data.ctypes.data), C.c_uint64(len(raw)), None, None, None, None, C.c_uint64(0))
        starts, ends, be = (np.zeros(max(int(n), 1), np.uint64) for _ in range(3))
        valid = np.zeros(max(int(n), 1), np.uint8)
        self.lib.fgo_frame(C.c_int(fr), vp(data.ctypes.data), C.c_uint64(len(raw)), vp(starts.ctypes.data), vp(ends.ctypes.data),
                           vp(be.ctypes.data), vp(valid.ctypes.data), C.c_uint64(int(n)))
        return [(int(starts[i]), int(ends[i]), raw[int(starts[i]):int(be[i])], bool(valid[i])) for i in range(int(n))]

    def frame_arrays(self, data: np.ndarray, framing: str):
        """fgo_frame over a large stream without Python objects per frame -> (starts u64[n], ends_with_terminator u64[n], valid u8[n])"""
        self.lib.fgo_frame.restype = C.c_int64
        fr = 1 if framing == "line" else 2
        vp = C.c_void_p
        n = int(self.lib.fgo_frame(C.c_int(fr), vp(data.ctypes.data), C.c_uint64(data.size), None, None, None, None, C.c_uint64(0)))
        starts, ends, be = (np.zeros(max(n, 1), np.uint64) for _ in range(3))
        valid = np.zeros(max(n, 1), np.uint8)
        self.lib.fgo_frame(C.c_int(fr), vp(data.ctypes.data), C.c_uint64(data.size), vp(starts.ctypes.data), vp(ends.ctypes.data),
                           vp(be.ctypes.data), vp(valid.ctypes.data), C.c_uint64(n))
        return starts[:n], ends[:n], valid[:n]

    def decode_stdout(self, fmt: int, line: bytes, config=None) -> bytes:
        """what Decoder::decode(line) prints to stdout (ltsv_decoder.rs:99)"""
        cfg, keep = self.make_cfg(config)
        cfgp = C.byref(cfg) if cfg is not None else None
        self.lib.fgo_decode_stdout.restype = C.c_int64
        self.lib.fgo_decode_stdout.argtypes = [C.c_int, C.c_void_p, C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64]
        n = self.lib.fgo_decode_stdout(fmt, cfgp, line, len(line), None, 0)
        buf = C.create_string_buffer(int(n) + 1)
        self.lib.fgo_decode_stdout(fmt, cfgp, line, len(line), buf, n)
        return buf.raw[:n]

    def bench(self, fmt: int, data: np.ndarray, offsets: np.ndarray, threads: int, config=None):
        cfg, keep = self.make_cfg(config)
        cfgp = C.byref(cfg) if cfg is not None else None
        chk, nok = C.c_uint64(), C.c_uint64()
        secs = self.lib.fgo_bench_decode(fmt, cfgp, data.ctypes.data, offsets.ctypes.data, len(offsets) - 1, threads,
                                         C.byref(chk), C.byref(nok))
        return secs, nok.value

    def bench_pipeline(self, fmt: int, enc: int, merger: int, data: np.ndarray, offsets: np.ndarray, threads: int, config=None):
        """decode -> encode -> merger -> null sink, threaded; -> (seconds, lines that reached the sink, encoded bytes)"""
        cfg, keep = self.make_cfg(config)
        cfgp = C.byref(cfg) if cfg is not None else None
        o, keep2 = self._opts(None, None, 0.0)
        nb, nok = C.c_uint64(), C.c_uint64()
        self.lib.fgo_bench_pipeline.restype = C.c_double
        secs = self.lib.fgo_bench_pipeline(C.c_int(fmt), cfgp, C.c_int(enc), C.c_int(merger), C.byref(o), C.c_void_p(data.ctypes.data),
                                           C.c_void_p(offsets.ctypes.data), C.c_uint64(len(offsets) - 1), C.c_int(threads),
                                           C.byref(nb), C.byref(nok))
        return secs, nok.value, nb.value

    def bench_timed(self, fmt: int, data: np.ndarray, offsets: np.ndarray, threads: int, seconds: float, config=None, enc: int = -1,
                    merger: int = 0):
        """fgo_bench_timed: persistent threads, each walking the whole tile for `seconds`; -> (wall seconds, lines handled)."""
        cfg, keep = self.make_cfg(config)
        cfgp = C.byref(cfg) if cfg is not None else None
        o, keep2 = self._opts(None, None, 0.0)
        lines, chk = C.c_uint64(), C.c_uint64()
        self.lib.fgo_bench_timed.restype = C.c_double
        secs = self.lib.fgo_bench_timed(C.c_int(fmt), cfgp, C.c_int(enc), C.c_int(merger), C.byref(o), C.c_void_p(data.ctypes.data),
                                        C.c_void_p(offsets.ctypes.data), C.c_uint64(len(offsets) - 1), C.c_int(threads),
                                        C.c_double(seconds), C.byref(lines), C.byref(chk))
        if secs <= 0:
            raise RuntimeError("fgo_bench_timed failed")
        return secs, lines.value

    def rfc3339(self, s: str):
        out = C.c_double()
        b = s.encode()
        return out.value if self.lib.fgo_rfc3339_to_unix(b, len(b), C.byref(out)) else None

    def parse_f64(self, s: str):
        out = C.c_double()
        b = s.encode()
        return out.value if self.lib.fgo_rust_parse_f64(b, len(b), C.byref(out)) else None

    def english(self, s: str):
        out = C.c_double()
        b = s.encode()
        return out.value if self.lib.fgo_english_time_to_unix(b, len(b), C.byref(out)) else None

    def json_number(self, s: str):
        kind, bits = C.c_int(), C.c_uint64()
        b = s.encode()
        if not self.lib.fgo_json_number(b, len(b), C.byref(kind), C.byref(bits)):
            return None
        return kind.value, bits.value
