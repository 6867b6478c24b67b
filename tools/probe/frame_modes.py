"""fg_frame_device on a resident stream: the one-pass chained scan against the classic three kernels (FG_LO_FRAME_CLASSIC), HIP events.
usage: python tools/probe/frame_modes.py [GiB = 4]"""
import sys, time
import numpy as np
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from flowgger_amd import RFC5424Decoder, synth
from flowgger_amd import _lib as L
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
dec = RFC5424Decoder()
dev = torch.device("cuda", dec.device)
lines = synth.rfc5424_lines(250000, cfg=2)
one = torch.frombuffer(bytearray(b"\n".join(lines) + b"\n"), dtype=torch.uint8).to(dev)
reps = max(1, int(gib * (1 << 30) / one.numel()))
d = torch.cat([one.repeat(reps), torch.zeros(32, dtype=torch.uint8, device=dev)])
raw = d[:one.numel() * reps]
n_lines = len(lines) * reps
for classic in (False, True, False, True):
    dec.set_launch_opts(frame_classic=classic)
    d_off, d_bad, n = dec.frame_device(raw, L.FG_FRAME_LINE, cap_frames=n_lines + 16)
    assert n == n_lines
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(3)]
    for a, b in ev:
        a.record()
        dec.frame_device(raw, L.FG_FRAME_LINE, cap_frames=n_lines + 16)
        b.record()
    torch.cuda.synchronize()
    ms = min(a.elapsed_time(b) for a, b in ev)
    print("classic " if classic else "one-pass", round(ms, 3), "ms", round(raw.numel() / ms / 1e6, 1), "GB/s", flush=True)
