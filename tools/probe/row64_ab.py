#!/usr/bin/env python3
"""The headline kernel with the fixed part of a line written as nine column stores (the library) or as ONE 64-byte record (the
measurement variants libfg_hip_row64a.so: four 16-byte stores per lane; libfg_hip_row64b.so: the wave's records transposed through
LDS, four stores of 1 KiB): VERDICT r5 item 8.  One process per library (FLOWGGER_AMD_LIB), the same 100 M-line cfg2 batch (a tile of
250 000 lines replicated 400 times), HIP events around each launch, median and best of 12 after 3 warm-ups.  The variants' tables are
not the ABI's -- only the time of the launch is compared."""
import json
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from flowgger_amd import RFC5424Decoder, synth  # noqa: E402
from flowgger_amd.tables import DeviceTables  # noqa: E402


def main():
    tile, reps = int(os.environ.get("FG_PROBE_TILE", "250000")), int(os.environ.get("FG_PROBE_REPS", "400"))
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream(dev)
    dec = RFC5424Decoder()
    lines = synth.rfc5424_lines(tile, cfg=2)
    data, offsets = synth.pack(lines)
    tb, n = int(offsets[-1]), tile * reps
    d_bytes = torch.from_numpy(data[:tb].copy()).to(dev).repeat(reps)
    d_bytes = torch.cat([d_bytes, torch.zeros(64, dtype=torch.uint8, device=dev)])
    off = torch.from_numpy(offsets[:-1].astype(np.int64)).to(dev)
    d_off = (off.unsqueeze(0) + (torch.arange(reps, device=dev, dtype=torch.int64) * tb).unsqueeze(1)).reshape(-1)
    d_off = torch.cat([d_off, torch.tensor([tb * reps], device=dev, dtype=torch.int64)])
    tables = DeviceTables(n + 64, 4096, dev)

    def call():
        dec.decode_device(d_bytes, d_off, tables, stream=stream)

    for _ in range(3):
        call()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(12)]
    for a, b in ev:
        a.record(stream)
        call()
        b.record(stream)
    torch.cuda.synchronize(dev)
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    alg = tb * reps + 8 * n + 68 * n
    print(json.dumps({"lib": os.environ.get("FLOWGGER_AMD_LIB", "libfg_hip.so"), "lines": n, "median_ms": round(ts[len(ts) // 2], 4), "best_ms": round(ts[0], 4),
                      "G_lines_s": round(n / ts[len(ts) // 2] / 1e6, 3), "frac_of_8TBps_by_the_ABI_bytes": round(alg / (ts[len(ts) // 2] * 1e-3) / 8e12, 4)}), flush=True)


if __name__ == "__main__":
    main()
