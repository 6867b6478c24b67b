#!/usr/bin/env python3
"""What an RFC3164 group costs by LINE SHAPE: tools/probe/rfc3164_shapes.py
The corpus of bench.py's rfc3164 workload mixes three shapes -- "<pri>Mon dd hh:mm:ss host app[pid]: msg", the same with a zone name
behind the time, and the custom form "<pri>host: yyyy Mon dd hh:mm:ss: app: msg" -- and a wave walks every shape one of its lanes
holds.  Batches of ONE shape each (and the mix) through fg_decode_batch_device, kernel-only, at 64 K, 1 M and 16 M lines (1 M lines of
222 bytes fit the 256 MB memory-side cache: only the last size is an HBM figure), lines in arrival order (`plain kernel`) and -- from 262 144
lines on, the library's own choice -- regrouped by shape (round 6)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch  # noqa: E402

import bench  # noqa: E402
from flowgger_amd import synth  # noqa: E402

ZONES = (b"America/", b"Europe/", b"Asia/", b"UTC", b"Africa/", b"Australia/", b"Pacific/", b"Etc/", b"GMT")


def shape(ln):
    head = ln[:64]
    if b": 20" in head or b": 19" in head:
        return "custom"
    return "zone" if any(z in head for z in ZONES) else "plain"


def main():
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream(dev)
    lines = synth.rfc3164_lines(400_000, invalid_frac=0.0)
    groups = {"mix": lines}
    for ln in lines:
        groups.setdefault(shape(ln), []).append(ln)
    for name, ls in groups.items():
        for n, mode in ((65536, 2), (65536, 0), (1 << 20, 2), (1 << 20, 0), (1 << 24, 2), (1 << 24, 1), (1 << 24, 0)):
            tile = ls[: min(len(ls), 65536)]
            reps = max(1, n // len(tile))
            R = bench.Resident(3, tile, reps, dev, 0, {"rfc3164_regroup": mode}, entries=False)
            for _ in range(3):
                R.decode(stream)
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(7)]
            for a, b in ev:
                a.record(stream)
                R.decode(stream)
                b.record(stream)
            torch.cuda.synchronize(dev)
            ts = sorted(a.elapsed_time(b) for a, b in ev)
            print(f"rfc3164 {name:7s} share {len(ls) / len(lines):5.2f}  n={R.n:8d}  {'plain kernel' if mode == 2 else 'global lists' if mode == 1 else 'library choice'}  {ts[3] * 1e3:8.1f} us  {R.n / ts[3] / 1e3:8.1f} M lines/s", flush=True)
            del R
            torch.cuda.empty_cache()


main()
