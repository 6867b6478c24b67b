#!/usr/bin/env python3
"""One corpus, many launch geometries: tools/sweep.py <workload> [--lines N] [--reps R[,R...]] opts;opts;...   (opts = k=v,k=v; empty = defaults)
Generates the workload's tile ONCE, keeps it resident, and times fg_decode_batch_device under each fg_set_launch_opts setting (HIP
events, 5 launches after 2 warm-ups) -- an A/B on the SAME box in seconds instead of one bench.py process per point."""
import os
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

import bench  # noqa: E402
from flowgger_amd import synth  # noqa: E402


def main():
    wl = sys.argv[1]
    args = sys.argv[2:]
    lines_n, reps_list = 1_000_000, [4]
    while args and args[0].startswith("--"):
        if args[0] == "--lines":
            lines_n = int(args[1])
        elif args[0] == "--reps":
            reps_list = [int(x) for x in args[1].split(",")]
        args = args[2:]
    settings = (args[0] if args else "").split(";")
    fmt = bench.WORKLOADS[wl][0]
    inv = float(os.environ.get("FG_SWEEP_INVALID", "0.01"))  # share of invalid lines in the corpus (the BASELINE corpora: 1 %)

    def gen():
        if wl == "cfg3":
            return synth.gelf_lines(lines_n, invalid_frac=inv)
        if wl in ("ltsv", "ltsv5"):
            return synth.ltsv_lines(lines_n, invalid_frac=inv, long_tail=wl == "ltsv5")
        if wl == "cfg5":
            return synth.rfc5424_lines(lines_n, cfg=5, sd=True, invalid_frac=inv, long_tail=True)
        return synth.rfc5424_lines(lines_n, cfg=4 if wl == "cfg4" else 2, sd=wl == "cfg4", invalid_frac=inv)

    lines = bench.cached_lines(f"sweep_{wl}_{lines_n}_{inv:g}", gen)  # (FG_BENCH_CACHE=<dir>: one pickle per corpus, for scripts that call this file repeatedly)
    dev = torch.device("cuda", 0)
    for reps in reps_list:
        sweep(wl, fmt, lines, reps, dev, settings)


def sweep(wl, fmt, lines, reps, dev, settings):
    R = bench.Resident(fmt, lines, reps, dev, 0, {}, entries=wl != "cfg2")
    stream = torch.cuda.current_stream(dev)
    for s in settings:
        opts = {k: int(v) for k, v in (kv.split("=") for kv in s.split(",") if kv)}
        R.dec.set_launch_opts(**opts)
        for _ in range(2):
            R.decode(stream)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
        for a, b in ev:
            a.record(stream)
            R.decode(stream)
            b.record(stream)
        torch.cuda.synchronize(dev)
        ms = float(np.median([a.elapsed_time(b) for a, b in ev]))
        n_ok, used = R.check_replicas()
        print(f"{wl:6s} n={R.n:10d} {s or 'defaults':40s} {R.n / ms / 1e3:9.1f} M lines/s  {ms:8.3f} ms  ok/tile {n_ok}", flush=True)
    del R
    torch.cuda.empty_cache()


main()
