"""fg_frame_decode_batch and fg_decode_batch from pinned memory at several batch sizes: the per-call fixed cost (intercept) and the rate."""
import sys, time, ctypes as C, numpy as np
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from flowgger_amd import GelfDecoder, LTSVDecoder, synth
from flowgger_amd import _lib as L
lib = L.lib()
def pinned(n):
    p = C.c_void_p(); L.check(lib.fg_alloc_pinned(n, C.byref(p)), "pin")
    return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (n,)), p
for name in sys.argv[1].split(","):
    if name == "cfg3": dec, lines = GelfDecoder(), synth.gelf_lines(250000)
    elif name == "cfg3clean": dec, lines = GelfDecoder(), synth.gelf_lines(250000, invalid_frac=0)
    else: dec, lines = LTSVDecoder(synth.LTSV_CONFIG), synth.ltsv_lines(250000)
    lines = [ln for ln in lines if b"\n" not in ln]
    one = np.frombuffer(b"\n".join(lines) + b"\n", np.uint8)
    reps = 16
    buf, h = pinned(one.size * reps + 64)
    for r in range(reps): buf[r*one.size:(r+1)*one.size] = one
    for r in (1, 2, 4, 8, 16):
        nb, n = one.size * r, len(lines) * r
        st, po, nf, cons = L.fg_tables(), C.c_void_p(), C.c_uint64(), C.c_uint64()
        def call():
            L.check(lib.fg_frame_decode_batch(dec._ctx, dec.fmt, 1, buf.ctypes.data, nb, 1, C.byref(st), C.byref(po), C.byref(nf), C.byref(cons)), "fdb")
        call(); call(); assert nf.value == n
        t0 = time.perf_counter()
        for _ in range(5): call()
        dt = (time.perf_counter() - t0) / 5
        print(name, n, "lines", round(nb / 1e6), "MB:", round(dt * 1e3, 2), "ms", round(n / dt / 1e6, 1), "M lines/s", flush=True)
    lib.fg_free_pinned(h)
