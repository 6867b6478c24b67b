"""ctypes binding of libfg_hip.so (include/fg_hip.h).  Fails loudly: no library -> ImportError-like
RuntimeError, no gfx950 device -> FgError(FG_ERR_NO_DEVICE).  There is no CPU fallback."""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import os

_HERE = Path(__file__).resolve().parent
# FLOWGGER_AMD_PROF_LIB=1 (tools/ only): the measurement build, `FG_BUILD_PROF=1 python -m flowgger_amd.build` (s_memtime phase clocks,
# ablation flags -- csrc/fg_pipeline.hpp).  The product library has neither.
# FLOWGGER_AMD_LIB=<bare file name beside this module> (tools/ only): another build of the same ABI, e.g. a branch's kernels for an A/B on
# one box.  A BARE NAME only -- anything with a directory part is refused, so the variable cannot point the product package at an
# arbitrary shared object -- and fg_abi_version() must match.  This is the PYTHON loader's choice of file; the library itself
# reads no environment variable.
_ALT = os.environ.get("FLOWGGER_AMD_LIB") or ""
if _ALT and (Path(_ALT).name != _ALT or not _ALT.startswith("libfg_hip") or not _ALT.endswith(".so")):
    raise RuntimeError(f"FLOWGGER_AMD_LIB={_ALT!r}: only a bare libfg_hip*.so file name beside flowgger_amd/_lib.py is accepted")
LIB_PATH = _HERE / (_ALT or ("libfg_hip_prof.so" if os.environ.get("FLOWGGER_AMD_PROF_LIB") else "libfg_hip.so"))

FG_ABI_VERSION = 4  # include/fg_hip.h
FG_RFC5424, FG_LTSV, FG_GELF, FG_RFC3164 = 0, 1, 2, 3
FG_FRAME_NONE, FG_FRAME_LINE, FG_FRAME_NUL = 0, 1, 2
FG_ST_OVERFLOW, FG_ST_BAD_UTF8 = 0xFE, 0xFD
FG_F_LTSV_NOVALUE = 128  # meta flags bit (include/fg_hip.h)
FG_OK, FG_ERR_ARG, FG_ERR_HIP, FG_ERR_NO_DEVICE, FG_ERR_ENT_OVERFLOW, FG_ERR_UNSUPPORTED, FG_ERR_NOMEM = 0, -1, -2, -3, -4, -5, -6
FG_YEAR_NOW = -2147483648  # INT32_MIN (include/fg_hip.h)
FG_NONE = 0xFFFFFFFF
FG_T_STRING, FG_T_BOOL, FG_T_F64, FG_T_I64, FG_T_U64, FG_T_NULL, FG_T_SDID = range(7)
FG_TABLE_ARRAYS = 15
_ERRNAMES = {-1: "FG_ERR_ARG", -2: "FG_ERR_HIP", -3: "FG_ERR_NO_DEVICE (no gfx950 GPU; there is no CPU fallback)",
             -4: "FG_ERR_ENT_OVERFLOW", -5: "FG_ERR_UNSUPPORTED", -6: "FG_ERR_NOMEM"}


class FgError(RuntimeError):
    def __init__(self, code: int, what: str = ""):
        self.code = code
        super().__init__(f"{what}: {_ERRNAMES.get(code, code)}")


class fg_span(C.Structure):
    _fields_ = [("off", C.c_uint32), ("len", C.c_uint32)]


class fg_tables(C.Structure):
    _fields_ = [
        ("n", C.c_uint64), ("ent_cap", C.c_uint64),
        ("meta", C.c_void_p), ("ts", C.c_void_p),
        ("hostname", C.c_void_p), ("appname", C.c_void_p), ("procid", C.c_void_p),
        ("msgid", C.c_void_p), ("msg", C.c_void_p), ("full_msg", C.c_void_p),
        ("ent_first", C.c_void_p), ("ent_count", C.c_void_p),
        ("ent_name", C.c_void_p), ("ent_val", C.c_void_p),
        ("ent_type", C.c_void_p), ("ent_flags", C.c_void_p), ("ent_used", C.c_void_p),
    ]


TABLE_FIELDS = ["meta", "ts", "hostname", "appname", "procid", "msgid", "msg", "full_msg", "ent_first",
                "ent_count", "ent_name", "ent_val", "ent_type", "ent_flags", "ent_used"]


class fg_gelf_extra(C.Structure):
    _fields_ = [("n", C.c_uint32), ("keys", C.POINTER(C.c_char_p)), ("values", C.POINTER(C.c_char_p))]


FG_ENC_GELF, FG_ENC_LTSV, FG_ENC_RFC5424, FG_ENC_RFC3164, FG_ENC_PASSTHROUGH = range(5)
FG_MERGE_NONE, FG_MERGE_LINE, FG_MERGE_NUL, FG_MERGE_SYSLEN = range(4)


class fg_encode_cfg(C.Structure):
    _fields_ = [("encoder", C.c_int), ("merger", C.c_int), ("n_extra", C.c_uint32), ("extra_keys", C.POINTER(C.c_char_p)),
                ("extra_values", C.POINTER(C.c_char_p)), ("prepend", C.c_char_p), ("now_ts", C.c_double)]


class fg_launch_opts(C.Structure):
    _fields_ = [("lines_per_group", C.c_uint32), ("tile_cap", C.c_uint32), ("waves_per_cu", C.c_uint32),
                ("gelf_lds_budget", C.c_uint32), ("gelf_window_kib", C.c_uint32), ("flags", C.c_uint32), ("chunk_lines", C.c_uint32), ("ent_chunk", C.c_uint32), ("fused_look", C.c_uint32), ("fused_ext", C.c_uint32)]


FG_LO_GELF_GENERIC, FG_LO_TRANSCODE_ONE_PIECE, FG_LO_NO_HEAD, FG_LO_FORCE_HEAD, FG_LO_SD_WALK, FG_LO_SD_PAIRS, FG_LO_NO_ZERO_COPY, FG_LO_FRAME_KERNEL_UPLOAD = 1, 2, 4, 8, 16, 32, 64, 128
FG_LO_FRAME_CLASSIC = 256
FG_LO_STATIC_CHUNKS = 512
FG_LO_FRAME_SELFTEST_STALL = 1024
FG_LO_NO_TAPER = 2048
FG_LO_TAPER_1 = 4096
FG_LO_TAPER_2 = 8192
FG_LO_NO_FUSED_FRAMING = 16384
FG_LO_RFC3164_REGROUP, FG_LO_RFC3164_NO_REGROUP = 32768, 65536
FG_PATH_DECODE_ZERO_COPY, FG_PATH_DECODE_SLICED, FG_PATH_FRAME_FUSED, FG_PATH_FRAME_SLICED, FG_PATH_FRAME_ONE_PIECE = 1, 2, 3, 4, 5
FG_LO_RESERVED = 0x40000000  # the library's own (fg_set_launch_opts clears it)


class fg_transcoded(C.Structure):
    _fields_ = [("out", C.c_void_p), ("out_bytes", C.c_uint64), ("out_offsets", C.c_void_p), ("meta", C.c_void_p),
                ("enc_status", C.c_void_p), ("frame_offsets", C.c_void_p), ("n", C.c_uint64), ("consumed", C.c_uint64)]


class fg_tz_table(C.Structure):
    _fields_ = [("n_zones", C.c_uint32), ("names", C.POINTER(C.c_char_p)), ("zone_first", C.c_void_p), ("utc_start", C.c_void_p),
                ("utc_offset", C.c_void_p)]


class fg_rfc3164_cfg(C.Structure):
    _fields_ = [("current_year", C.c_int32), ("tz", C.POINTER(fg_tz_table))]


class fg_cfg(C.Structure):
    _fields_ = [
        ("n_schema", C.c_uint32),
        ("schema_names", C.POINTER(C.c_char_p)),
        ("schema_types", C.POINTER(C.c_uint8)),
        ("suffix_bool", C.c_char_p), ("suffix_f64", C.c_char_p),
        ("suffix_i64", C.c_char_p), ("suffix_u64", C.c_char_p),
    ]


_lib = None


def lib() -> C.CDLL:
    """Load libfg_hip.so (after torch, when torch is importable, so both share one HIP runtime)."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m flowgger_amd.build` "
            "(hipcc --offload-arch=gfx950). The decoders have no CPU fallback.")
    try:  # make torch's HIP runtime the process-wide one before ours is resolved
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional for the library itself
        pass
    L = C.CDLL(str(LIB_PATH))
    vp, u64, u32 = C.c_void_p, C.c_uint64, C.c_uint32
    L.fg_abi_version.restype = C.c_int
    L.fg_create.argtypes = [C.c_int, C.POINTER(fg_cfg), C.POINTER(vp)]
    L.fg_clone.argtypes = [vp, C.POINTER(vp)]
    L.fg_destroy.argtypes = [vp]
    L.fg_destroy.restype = None
    L.fg_last_hip_error.argtypes = [vp]
    L.fg_set_launch_opts.argtypes = [vp, C.POINTER(fg_launch_opts)]
    L.fg_tables_layout.argtypes = [u64, u64, C.POINTER(u64)]
    L.fg_decode_batch_device.argtypes = [vp, C.c_int, vp, u64, vp, u64, C.POINTER(fg_tables), vp]
    L.fg_decode_batch.argtypes = [vp, C.c_int, vp, u64, vp, u64, C.POINTER(fg_tables)]
    L.fg_error_string.argtypes = [C.c_int, C.c_uint8]
    L.fg_error_string.restype = C.c_char_p
    L.fg_tables_serialize.argtypes = [C.c_int, C.POINTER(fg_cfg), vp, vp, C.POINTER(fg_tables), u64, u64, vp, u64, vp]
    L.fg_tables_serialize.restype = C.c_int64
    L.fg_tables_stdout.argtypes = [C.c_int, C.c_int, vp, vp, C.POINTER(fg_tables), u64, u64, vp, u64]
    L.fg_tables_stdout.restype = C.c_int64
    L.fg_shard_plan.argtypes = [vp, u64, u32, vp]
    L.fg_gather_size.argtypes = [C.POINTER(fg_tables), u32, C.POINTER(u64), C.POINTER(u64)]
    L.fg_gather_tables.argtypes = [C.POINTER(fg_tables), u32, C.POINTER(fg_tables)]
    L.fg_merge_tables.argtypes = [C.POINTER(fg_tables), u32, C.POINTER(vp), C.POINTER(fg_tables), vp]
    L.fg_merge_tables_device.argtypes = [vp, C.POINTER(fg_tables), u32, C.POINTER(vp), C.POINTER(fg_tables), vp, vp]
    L.fg_ordered_merge.argtypes = [u32, vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), vp, u64, vp]
    L.fg_ordered_merge.restype = C.c_int64
    L.fg_ticket_ring_check.argtypes = [vp]
    L.fg_set_pinned_limits.argtypes = [u64, u64]
    L.fg_pinned_stats.argtypes = [C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]
    L.fg_set_timing.argtypes = [vp, C.c_int]
    L.fg_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.fg_frame_decode_batch.argtypes = [vp, C.c_int, C.c_int, vp, u64, C.c_int, C.POINTER(fg_tables), C.POINTER(vp),
                                        C.POINTER(u64), C.POINTER(u64)]
    L.fg_encode_gelf_device.argtypes = [vp, C.c_int, vp, u64, vp, u64, C.POINTER(fg_tables), C.POINTER(fg_gelf_extra), vp, u64, vp,
                                        C.POINTER(u64), vp]
    L.fg_encode_device.argtypes = [vp, C.c_int, C.POINTER(fg_encode_cfg), vp, u64, vp, u64, C.POINTER(fg_tables), vp, u64, vp, vp,
                                   C.POINTER(u64), vp]
    L.fg_encode_device_async.argtypes = [vp, C.c_int, C.POINTER(fg_encode_cfg), vp, u64, vp, u64, C.POINTER(fg_tables), vp, u64, vp, vp,
                                         u64, vp]
    L.fg_transcode_batch.argtypes = [vp, C.c_int, C.c_int, C.POINTER(fg_encode_cfg), vp, u64, vp, u64, C.c_int, C.POINTER(fg_transcoded)]
    L.fg_encode_error_string.argtypes = [C.c_uint8]
    L.fg_encode_error_string.restype = C.c_char_p
    L.fg_set_rfc3164.argtypes = [vp, C.POINTER(fg_rfc3164_cfg)]
    L.fg_measure_link.argtypes = [vp, u64, C.POINTER(C.c_double)]
    L.fg_calibrate_device.argtypes = [vp, C.c_int, vp, vp, u64, vp]
    L.fg_alloc_pinned.argtypes = [u64, C.POINTER(vp)]
    L.fg_free_pinned.argtypes = [vp]
    L.fg_free_pinned.restype = None
    L.fg_frame_device.argtypes = [vp, C.c_int, vp, u64, vp, vp, u64, C.POINTER(u64), vp]
    L.fg_decode_frames_device.argtypes = [vp, C.c_int, C.c_int, vp, u64, vp, u64, vp, C.POINTER(fg_tables), vp]
    L.fg_last_host_path.argtypes = [vp]
    L.fg_frame_decode_device.argtypes = [vp, C.c_int, C.c_int, vp, u64, C.c_int, vp, u64, C.POINTER(fg_tables), u64, vp, vp]
    if L.fg_abi_version() != FG_ABI_VERSION:
        raise RuntimeError("libfg_hip.so ABI version mismatch")
    _lib = L
    return L


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise FgError(rc, what)
