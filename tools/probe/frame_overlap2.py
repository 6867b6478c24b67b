"""How many waves per CU should the decode grids of the host pipelines take?  fg_frame_decode_batch (copy-engine uploads, tables written
into pinned memory by the kernels) and fg_decode_batch (zero copy both ways), GELF / SD / LTSV / RFC5424."""
import sys, time, ctypes as C, numpy as np
sys.path.insert(0, '.')
import torch
from flowgger_amd import GelfDecoder, LTSVDecoder, RFC5424Decoder, synth
from flowgger_amd import _lib as L
lib = L.lib()
def pinned(n):
    p = C.c_void_p(); L.check(lib.fg_alloc_pinned(n, C.byref(p)), "pin")
    return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (n,)), p
for name in sys.argv[1].split(","):
    if name == "cfg3": dec, lines = GelfDecoder(), synth.gelf_lines(250000)
    elif name == "ltsv": dec, lines = LTSVDecoder(synth.LTSV_CONFIG), synth.ltsv_lines(250000)
    elif name == "cfg4": dec, lines = RFC5424Decoder(), synth.rfc5424_lines(250000, cfg=4, sd=True)
    else: dec, lines = RFC5424Decoder(), synth.rfc5424_lines(250000, cfg=2)
    lines = [ln for ln in lines if b"\n" not in ln]
    one = np.frombuffer(b"\n".join(lines) + b"\n", np.uint8)
    reps = 16
    buf, h = pinned(one.size * reps + 64)
    for r in range(reps): buf[r*one.size:(r+1)*one.size] = one
    n = len(lines) * reps
    # the framed form of the same batch, pinned too (fg_decode_batch)
    lens = np.fromiter((len(b) for b in lines), np.int64, len(lines))
    off1 = np.zeros(len(lines) + 1, np.uint64); off1[1:] = np.cumsum(lens)
    packed = np.frombuffer(b"".join(lines), np.uint8)
    pb, hb = pinned(packed.size * reps + 64)
    po_, ho = pinned((n + 1) * 8)
    offs = po_.view(np.uint64)
    for r in range(reps):
        pb[r*packed.size:(r+1)*packed.size] = packed
        offs[r*len(lines):(r+1)*len(lines)] = off1[:-1] + np.uint64(r * packed.size)
    offs[n] = np.uint64(reps * packed.size)
    for w in [int(x) for x in sys.argv[2].split(",")]:
        dec.set_launch_opts(waves_per_cu=w)
        st, po, nf, cons = L.fg_tables(), C.c_void_p(), C.c_uint64(), C.c_uint64()
        def call():
            L.check(lib.fg_frame_decode_batch(dec._ctx, dec.fmt, 1, buf.ctypes.data, one.size*reps, 1, C.byref(st), C.byref(po), C.byref(nf), C.byref(cons)), "fdb")
        call(); assert nf.value == n
        t0 = time.perf_counter()
        for _ in range(3): call()
        dt = (time.perf_counter() - t0) / 3
        def call2():
            L.check(lib.fg_decode_batch(dec._ctx, dec.fmt, pb.ctypes.data, packed.size*reps, offs.ctypes.data, n, C.byref(st)), "db")
        call2()
        t0 = time.perf_counter()
        for _ in range(3): call2()
        dt2 = (time.perf_counter() - t0) / 3
        print(name, "waves/CU", w or "all", "| frame_decode_batch", round(n/dt/1e6,1), "M lines/s | decode_batch", round(n/dt2/1e6,1), "M lines/s", flush=True)
    lib.fg_free_pinned(h); lib.fg_free_pinned(hb); lib.fg_free_pinned(ho)
