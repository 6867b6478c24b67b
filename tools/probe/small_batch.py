#!/usr/bin/env python3
"""Kernel-only decode time against the batch size: tools/probe/small_batch.py [workload ...]
For every workload the largest tile is generated once; a batch of n lines is its first n lines, resident in HBM, decoded by
fg_decode_batch_device (HIP events on the launch stream, median of 9 after 3 warm-ups).  The real caller (a framer) hands over
10^4 .. 10^6 lines: a fixed cost that 100 M-line runs hide shows here as  t = a + b * n.
FG_PROBE_OPTS='k=v,k=v;...' adds launch-geometry variants per size; FG_PROBE_FILTER=sd900 drops the lines whose structured data
ends behind byte 900 (the lines a 1 KiB head cannot hold)."""
import json
import os
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch  # noqa: E402

import bench  # noqa: E402
from flowgger_amd import synth  # noqa: E402

SIZES = [int(x) for x in os.environ.get("FG_PROBE_SIZES", "16384,65536,262144,524288,1048576,2097152").split(",")]


def gen(wl, n, invalid_frac):
    if wl == "cfg3":
        return synth.gelf_lines(n, invalid_frac=invalid_frac)
    if wl in ("ltsv", "ltsv5"):
        return synth.ltsv_lines(n, invalid_frac=invalid_frac, long_tail=wl == "ltsv5")
    if wl == "cfg5":
        return synth.rfc5424_lines(n, cfg=5, sd=True, invalid_frac=invalid_frac, long_tail=True)
    if wl == "rfc3164":
        return synth.rfc3164_lines(n, invalid_frac=invalid_frac)
    return synth.rfc5424_lines(n, cfg=4 if wl == "cfg4" else 2, sd=wl == "cfg4", invalid_frac=invalid_frac)


def time_decode(R, stream, dev, reps=9, warm=3):
    for _ in range(warm):
        R.decode(stream)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(stream)
        R.decode(stream)
        b.record(stream)
    torch.cuda.synchronize(dev)
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2], ts[0]


def main():
    wls = sys.argv[1:] or ["cfg2", "cfg5", "ltsv5", "cfg4", "cfg3", "ltsv"]
    invalid_frac = float(os.environ.get("FG_PROBE_INVALID", "0.01"))
    flt = os.environ.get("FG_PROBE_FILTER", "")
    variants = [v for v in os.environ.get("FG_PROBE_OPTS", "").split(";")]
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream(dev)
    out = {}
    for wl in wls:
        fmt = bench.WORKLOADS[wl][0]
        top = max(SIZES)
        if wl != "cfg2":  # (the entry corpora take Python a minute per million lines to format)
            top = min(top, int(os.environ.get("FG_PROBE_TOP", "524288")))
        key = f"probe_{wl}_{top}_{invalid_frac:g}"
        lines = bench.cached_lines(key, lambda: gen(wl, top, invalid_frac))
        if flt == "sd900":
            def sd_end(b):
                i = b.find(b"[")
                return 0 if i < 0 else b.rfind(b"] ") + 1
            before = len(lines)
            lines = [ln for ln in lines if sd_end(ln) <= 900]
            print(f"# {wl}: filter sd900 dropped {before - len(lines)} of {before} lines", flush=True)
        rows = []
        for n in SIZES:
            if n > len(lines):
                continue
            R = bench.Resident(fmt, lines[:n], 1, dev, 0, {}, entries=wl != "cfg2")
            for v in variants:
                opts = {k: int(x) for k, x in (kv.split("=") for kv in v.split(",") if kv)}
                R.dec.set_launch_opts(**opts)
                med, best = time_decode(R, stream, dev)
                R.check_replicas()
                rows.append({"n": n, "opts": v, "ms": med, "ms_min": best, "lines_per_s": n / (med * 1e-3)})
                print(f"{wl:6s} n={n:8d} {v or 'defaults':28s} {med * 1e3:9.1f} us (min {best * 1e3:8.1f})  {n / med / 1e3:9.1f} M lines/s", flush=True)
            del R
            torch.cuda.empty_cache()
        out[wl] = rows
    print(json.dumps(out))


main()
