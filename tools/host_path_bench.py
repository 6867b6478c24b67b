#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point fg_decode_batch (H2D + kernel + D2H of the tables).
Never the bench metric (bench.py times HBM-resident batches); reported in DESIGN.md next to it."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from flowgger_amd import RFC5424Decoder, synth  # noqa: E402

if "--workload" in sys.argv and sys.argv[sys.argv.index("--workload") + 1] == "latency":
    # the per-record callers (udp_input.rs:139, redis_input.rs:159, file/worker.rs:116) call decode() once per
    # record = a batch of one.  Sweep of the batch size through fg_decode_batch (host buffers, pinned): microseconds per call and
    # lines/s -- where the GPU path overtakes one CPU thread (BENCH cpu_baseline.single_thread, ~8 M lines/s = 0.12 us per line).
    import ctypes as C
    import json

    import numpy as np

    from flowgger_amd import _lib as L

    lines = synth.rfc5424_lines(4_000_000 if "--full" in sys.argv else 1_000_000, cfg=2)
    data, offsets = synth.pack(lines)
    dec = RFC5424Decoder()

    def pinned_copy(a):
        p = C.c_void_p()
        L.check(L.lib().fg_alloc_pinned(a.nbytes + 32, C.byref(p)), "fg_alloc_pinned")
        buf = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (a.nbytes,)).view(a.dtype)
        buf[:] = a
        return buf

    pdata, poffs = pinned_copy(data), pinned_copy(offsets)
    rows = []
    for b in (1, 8, 64, 512, 1024, 4096, 16384, 65536, 262144, len(lines)):
        if b > len(lines):
            continue
        st = L.fg_tables()
        nb = int(offsets[b])
        reps = max(3, min(2000, int(2_000_000 / b)))

        def call():
            L.check(L.lib().fg_decode_batch(dec._ctx, dec.fmt, pdata.ctypes.data, nb, poffs.ctypes.data, b, C.byref(st)), "fg_decode_batch")

        for _ in range(3):
            call()
        t0 = time.perf_counter()
        for _ in range(reps):
            call()
        dt = (time.perf_counter() - t0) / reps
        rows.append({"batch_lines": b, "us_per_call": dt * 1e6, "us_per_line": dt * 1e6 / b, "lines_per_s": b / dt})
        print(f"batch {b:>8} lines: {dt * 1e6:10.1f} us per call, {dt * 1e6 / b:9.3f} us per line, {b / dt / 1e6:9.2f} M lines/s", flush=True)
    print(json.dumps({"latency_sweep": rows, "entry": "fg_decode_batch (pinned host buffers in, tables in the ctx's pinned buffer out)"}))
    raise SystemExit(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 1_000_000
lines = synth.rfc5424_lines(n, cfg=2)
data, offsets = synth.pack(lines)
import ctypes as C

import numpy as np

from flowgger_amd import _lib as L  # noqa: E402


def pinned_copy(a):
    p = C.c_void_p()
    L.check(L.lib().fg_alloc_pinned(a.nbytes, C.byref(p)), "fg_alloc_pinned")
    buf = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (a.nbytes,)).view(a.dtype)
    buf[:] = a
    return buf


dec = RFC5424Decoder()
dec.decode_packed(data[: 1 << 20], offsets[:1000])  # warm-up (context, stash, staging buffers)
best = 1e9
for _ in range(3):
    t0 = time.perf_counter()
    tab = dec.decode_packed(data, offsets)
    best = min(best, time.perf_counter() - t0)
print(f"fg_decode_batch (pageable source): {n} lines, {data.size / 1e6:.0f} MB in {best * 1e3:.1f} ms = {n / best / 1e6:.1f} M lines/s, "
      f"{data.size / best / 1e9:.2f} GB/s of input (includes the Python-side copy of the tables)")

pdata, poffs = pinned_copy(data), pinned_copy(offsets)
best = 1e9
for _ in range(3):
    t0 = time.perf_counter()
    st = L.fg_tables()
    L.check(L.lib().fg_decode_batch(dec._ctx, dec.fmt, pdata.ctypes.data, pdata.size, poffs.ctypes.data, n, C.byref(st)), "fg_decode_batch")
    best = min(best, time.perf_counter() - t0)
print(f"fg_decode_batch (pinned source, tables left in the ctx's pinned buffer): {best * 1e3:.1f} ms = {n / best / 1e6:.1f} M lines/s, "
      f"{data.size / best / 1e9:.2f} GB/s of input")

# fg_transcode_batch: chunk in -> decode -> GELF encode -> line merger -> bytes out (BASELINE configs[0] on the GPU)
from flowgger_amd import GelfEncoder  # noqa: E402

enc = GelfEncoder(None, merger="line")
cfg, _keep = enc._cfg_struct(0.0)
for label, framing, src, offs_ptr, nn in (
        ("framed lines (pinned)", L.FG_FRAME_NONE, pdata, poffs.ctypes.data, n),
        ("raw '\\n' stream framed on the GPU (pinned)", L.FG_FRAME_LINE, pinned_copy(np.frombuffer(b"\n".join(lines) + b"\n", np.uint8)), None, 0)):
    best, res = 1e9, L.fg_transcoded()
    for _ in range(4):
        t0 = time.perf_counter()
        L.check(L.lib().fg_transcode_batch(dec._ctx, dec.fmt, framing, C.byref(cfg), src.ctypes.data, src.size, offs_ptr, nn, 1, C.byref(res)),
                "fg_transcode_batch")
        best = min(best, time.perf_counter() - t0)
    print(f"fg_transcode_batch, {label}: {int(res.n)} lines, {src.size / 1e6:.0f} MB in, {int(res.out_bytes) / 1e6:.0f} MB out, "
          f"{best * 1e3:.1f} ms = {int(res.n) / best / 1e6:.1f} M lines/s ({(src.size + int(res.out_bytes)) / best / 1e9:.1f} GB/s over PCIe, both directions)")
