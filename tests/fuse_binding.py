"""ctypes binding of tests/native/libfuse_host.so: flowgger_amd/csrc/fg_fuse.hpp (framing inside the decode kernels) compiled for the
CPU over the fiber emulation of a wavefront (test infrastructure)."""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent / "native"
ROOT = HERE.parent.parent
LIB = HERE / "libfuse_host.so"
SRC = [HERE / "fuse_host.cpp", HERE / "fg_wave_emu.hpp", ROOT / "flowgger_amd" / "csrc" / "fg_fuse.hpp", ROOT / "flowgger_amd" / "csrc" / "fg_wave.hpp"]


def build() -> Path:
    if not LIB.exists() or any(s.exists() and s.stat().st_mtime > LIB.stat().st_mtime for s in SRC):
        subprocess.run(["g++", "-O2", "-g", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function", "-Wno-unknown-pragmas",
                        f"-I{ROOT / 'include'}", f"-I{HERE}", f"-I{ROOT / 'flowgger_amd' / 'csrc'}", "-o", str(LIB), str(HERE / "fuse_host.cpp")],
                       check=True)
    return LIB


class FuseHost:
    def __init__(self):
        self.lib = C.CDLL(str(build()))
        self.lib.fgf_last_error.restype = C.c_char_p
        self.lib.fgf_frame.restype = C.c_long
        self.lib.fgf_plan.restype = C.c_uint32

    def plan(self, avg_len: int, lines: int, tile_cap: int):
        look = C.c_uint32()
        s = self.lib.fgf_plan(C.c_uint64(avg_len), C.c_uint32(lines), C.c_uint32(tile_cap), C.byref(look))
        return int(s), int(look.value)

    def frame(self, raw: bytes, delim: int, final: bool, S: int, look: int, tile_cap: int, lines: int = 64, garbage: int = 0x0A):
        """-> (starts, ends, bad, consumed, passes, scans); what lies behind the stream's end inside its last 16 bytes is `garbage`"""
        n = len(raw)
        padded = (n + 15) & ~15
        buf = np.full(max(padded, 16), garbage, np.uint8)
        buf[:n] = np.frombuffer(raw, np.uint8)
        cap = n + 2
        starts, ends = np.zeros(cap, np.uint64), np.zeros(cap, np.uint64)
        bad = np.zeros(cap, np.uint8)
        consumed, passes, scans = C.c_uint64(), C.c_uint64(), C.c_uint64()
        r = self.lib.fgf_frame(C.c_void_p(buf.ctypes.data), C.c_uint64(n), C.c_uint32(delim), C.c_int(1 if final else 0), C.c_uint32(S),
                               C.c_uint32(look), C.c_uint32(tile_cap), C.c_uint32(lines), C.c_void_p(starts.ctypes.data),
                               C.c_void_p(ends.ctypes.data), C.c_void_p(bad.ctypes.data), C.c_uint64(cap), C.byref(consumed), C.byref(passes),
                               C.byref(scans))
        if r < 0:
            raise RuntimeError(self.lib.fgf_last_error().decode())
        return starts[:r], ends[:r], bad[:r], int(consumed.value), int(passes.value), int(scans.value)
