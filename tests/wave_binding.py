"""ctypes binding of tests/native/libwave_host.so: the wave-cooperative tokenisers compiled for the CPU over the fiber
emulation of a wavefront (test infrastructure)."""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

from flowgger_amd import _lib as L
from flowgger_amd.tables import HostTables, _DT, layout

HERE = Path(__file__).resolve().parent / "native"
ROOT = HERE.parent.parent
LIB = HERE / "libwave_host.so"
SRC = [HERE / "wave_host.cpp", HERE / "fg_wave_emu.hpp"] + sorted((ROOT / "flowgger_amd" / "csrc").glob("fg_*2.hpp")) + \
      [ROOT / "flowgger_amd" / "csrc" / n for n in ("fg_wave.hpp", "fg_numparse.hpp", "fg_timeconv.hpp", "fg_tables_view.hpp")] + \
      [ROOT / "include" / "fg_hip.h"]


def build() -> Path:
    if not LIB.exists() or any(s.exists() and s.stat().st_mtime > LIB.stat().st_mtime for s in SRC):
        subprocess.run(["g++", "-O2", "-g", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function", "-Wno-unknown-pragmas",
                        "-fno-fast-math", "-ffp-contract=off", f"-I{ROOT / 'include'}", f"-I{HERE}", f"-I{ROOT / 'flowgger_amd' / 'csrc'}",
                        "-o", str(LIB), str(HERE / "wave_host.cpp")], check=True)
    return LIB


def empty_tables(n: int, ent_cap: int) -> HostTables:
    offs, _ = layout(n, ent_cap)
    arrays = {}
    for name, (_, size) in zip(L.TABLE_FIELDS, offs):
        dt = np.dtype(_DT[name])
        arrays[name] = np.zeros(max(size // dt.itemsize, 1), dt)
    arrays["meta"][:] = 0xFC  # "not produced"
    return HostTables(n, ent_cap, arrays)


class WaveHost:
    def __init__(self):
        self.lib = C.CDLL(str(build()))
        self.lib.fgw_last_error.restype = C.c_char_p

    def gelf(self, data: np.ndarray, offsets: np.ndarray, lines_per_group=32, tile_cap=12288, strip=0, ent_cap=None):
        """-> (HostTables, handled uint8[n]).  strip = FG_FRAME_LINE (1) / FG_FRAME_NUL (2): `offsets` delimit frames of a raw stream
        including their terminators."""
        data = np.ascontiguousarray(data, np.uint8)
        offsets = np.ascontiguousarray(offsets, np.uint64)
        n = len(offsets) - 1
        t = empty_tables(n, int(data.size) // 8 + 1024 if ent_cap is None else ent_cap)
        handled = np.zeros(max(n, 1), np.uint8)
        rc = self.lib.fgw_gelf_decode_framed(C.c_void_p(data.ctypes.data), C.c_uint64(data.size), C.c_void_p(offsets.ctypes.data),
                                             C.c_uint64(n), C.byref(t.struct), C.c_uint32(lines_per_group), C.c_uint32(tile_cap),
                                             C.c_void_p(handled.ctypes.data), C.c_uint32(strip))
        if rc != 0:
            raise RuntimeError(self.lib.fgw_last_error().decode())
        return t, handled[:n]

