"""fg_frame_decode_batch, GELF / LTSV raw streams in pinned memory: copy-engine uploads vs the scan that reads the pinned chunk itself,
each with the decode grid at full occupancy and with wave slots left free for the next slice's framing kernels."""
import sys, time, ctypes as C, numpy as np
sys.path.insert(0, '.')
import torch
from flowgger_amd import GelfDecoder, LTSVDecoder, synth
from flowgger_amd import _lib as L
lib = L.lib()
def pinned(n):
    p = C.c_void_p(); L.check(lib.fg_alloc_pinned(n, C.byref(p)), "pin")
    return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (n,)), p
for name in ("cfg3", "ltsv"):
    if name == "cfg3": dec, lines = GelfDecoder(), synth.gelf_lines(250000)
    else: dec, lines = LTSVDecoder(synth.LTSV_CONFIG), synth.ltsv_lines(250000)
    lines = [ln for ln in lines if b"\n" not in ln]
    one = np.frombuffer(b"\n".join(lines) + b"\n", np.uint8)
    reps = 16
    buf, h = pinned(one.size * reps + 64)
    for r in range(reps): buf[r*one.size:(r+1)*one.size] = one
    n = len(lines) * reps
    for ku, w in ((False, 0), (False, 8), (False, 4), (True, 0), (True, 12), (True, 8), (True, 4), (True, 2)):
        dec.set_launch_opts(frame_kernel_upload=ku, waves_per_cu=w)
        st, po, nf, cons = L.fg_tables(), C.c_void_p(), C.c_uint64(), C.c_uint64()
        def call():
            L.check(lib.fg_frame_decode_batch(dec._ctx, dec.fmt, 1, buf.ctypes.data, one.size*reps, 1, C.byref(st), C.byref(po), C.byref(nf), C.byref(cons)), "fdb")
        call(); assert nf.value == n
        t0 = time.perf_counter()
        for _ in range(3): call()
        dt = (time.perf_counter() - t0) / 3
        print(name, "kernel_upload" if ku else "copy_engine  ", "waves/CU", w or "all", round(n/dt/1e6,1), "M lines/s", round(one.size*reps/dt/1e9,1), "GB/s in", flush=True)
    lib.fg_free_pinned(h)
