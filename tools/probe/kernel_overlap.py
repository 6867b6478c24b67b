"""Do two KERNELS on two streams run side by side on this platform, one reading pinned host memory (an upload by kernel) and one writing
pinned host memory (a download by kernel)?  fg_calibrate_device float4 copies: pinned -> HBM on stream A, HBM -> pinned on stream B,
each alone and both at once; and the same with hipMemcpyAsync for the upload (the copy engine)."""
import sys, time, ctypes as C
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np
import torch
from flowgger_amd import RFC5424Decoder
from flowgger_amd import _lib as L
lib = L.lib()
dec = RFC5424Decoder()
dev = torch.device("cuda", dec.device)
N = 512 << 20
def pinned(n):
    p = C.c_void_p(); L.check(lib.fg_alloc_pinned(n, C.byref(p)), "pin")
    return p
h_in, h_out = pinned(N), pinned(N)
np.ctypeslib.as_array(C.cast(h_in, C.POINTER(C.c_uint8)), (N,))[:] = 7
d_a = torch.empty(N, dtype=torch.uint8, device=dev)
d_b = torch.ones(N, dtype=torch.uint8, device=dev)
sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
def up_kernel():   L.check(lib.fg_calibrate_device(dec._ctx, 0, h_in, d_a.data_ptr(), N, C.c_void_p(sa.cuda_stream)), "up")
def down_kernel(): L.check(lib.fg_calibrate_device(dec._ctx, 0, d_b.data_ptr(), h_out, N, C.c_void_p(sb.cuda_stream)), "down")
hin_t = torch.empty(N, dtype=torch.uint8, pin_memory=True)
def up_engine():
    with torch.cuda.stream(sa): d_a.copy_(hin_t, non_blocking=True)
def timed(fs, reps=3):
    for f in fs: f()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(reps):
        for f in fs: f()
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / reps * 1e3
for name, fs in (("upload by kernel", [up_kernel]), ("download by kernel", [down_kernel]), ("both kernels", [up_kernel, down_kernel]),
                 ("upload by copy engine", [up_engine]), ("copy-engine upload + kernel download", [up_engine, down_kernel])):
    ms = timed(fs)
    print(f"{name:40s} {ms:7.2f} ms  {N * len(fs) / ms / 1e6:6.1f} GB/s total", flush=True)
