#!/usr/bin/env python3
"""Where the time of one fg_transcode_batch call goes: the driver's e2e sample -- 4 tiles of 250 000 cfg2 lines
from pinned memory, GELF text back -- called a few times with pauses between the calls, to be run under
  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d <dir> -o tt -- python tools/probe/transcode_timeline.py run
and then   python tools/probe/transcode_timeline.py report <dir>   turns the two traces of the LAST call into a timeline:
every copy and every kernel with start / end in ms from the call's first operation, the busy time of the two directions of the link,
how long both were busy at once, and the stretches in which the download direction (the bound: the text is 2.4x the input) idles."""
import csv
import glob
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))


def run():
    import ctypes as C

    import numpy as np
    import torch  # noqa: F401

    from flowgger_amd import GelfEncoder, RFC5424Decoder, synth
    from flowgger_amd import _lib as L

    tile, reps = int(sys.argv[4]) if len(sys.argv) > 4 else 250_000, int(sys.argv[2]) if len(sys.argv) > 2 else 4
    lines = synth.rfc5424_lines(tile, cfg=2)
    data, offsets = synth.pack(lines)
    tb, n = int(offsets[-1]), tile * reps
    lib = L.lib()
    dec = RFC5424Decoder()
    if len(sys.argv) > 3 and sys.argv[3]:  # launch options, k=v,k=v
        dec.set_launch_opts(**{k: int(v) for k, v in (kv.split("=") for kv in sys.argv[3].split(","))})

    def pinned(nbytes):
        p = C.c_void_p()
        L.check(lib.fg_alloc_pinned(nbytes, C.byref(p)), "fg_alloc_pinned")
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (nbytes,))

    pdata = pinned(tb * reps + 64)
    poffs = pinned((n + 1) * 8 + 64)[: (n + 1) * 8].view(np.uint64)
    for r in range(reps):
        pdata[r * tb:(r + 1) * tb] = data[:tb]
        poffs[r * tile:(r + 1) * tile] = offsets[:-1] + np.uint64(r * tb)
    poffs[n] = tb * reps
    gb = (C.c_double * 3)()
    lib.fg_measure_link(dec._ctx, 1 << 30, gb)
    enc = GelfEncoder(None, merger="line")
    cfg, _keep = enc._cfg_struct(0.0)
    res = L.fg_transcoded()
    walls = []
    for _ in range(4):
        time.sleep(0.25)  # (the pause is how `report` finds the calls in the traces)
        t0 = time.perf_counter()
        L.check(lib.fg_transcode_batch(dec._ctx, dec.fmt, L.FG_FRAME_NONE, C.byref(cfg), pdata.ctypes.data, tb * reps, poffs.ctypes.data, n, 1, C.byref(res)),
                "fg_transcode_batch")
        walls.append(time.perf_counter() - t0)
    time.sleep(0.25)
    print(json.dumps({"lines": n, "in_bytes": tb * reps + 8 * (n + 1), "out_bytes": int(res.out_bytes), "wall_ms": [round(w * 1e3, 3) for w in walls],
                      "link_GBps": [round(float(x), 2) for x in gb]}), flush=True)


def load(dirname, pat):
    rows = []
    for f in glob.glob(f"{dirname}/**/*{pat}.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    return rows


def union(iv):
    iv = sorted(iv)
    out = []
    for a, b in iv:
        if out and a <= out[-1][1]:
            out[-1][1] = max(out[-1][1], b)
        else:
            out.append([a, b])
    return out


def overlap(u1, u2):
    t, i, j = 0.0, 0, 0
    while i < len(u1) and j < len(u2):
        a, b = max(u1[i][0], u2[j][0]), min(u1[i][1], u2[j][1])
        if b > a:
            t += b - a
        if u1[i][1] < u2[j][1]:
            i += 1
        else:
            j += 1
    return t


def report(dirname, meta=None):
    ops = []
    for r in load(dirname, "kernel_trace"):
        nm = r["Kernel_Name"].split("(")[0][:60]
        # (the runtime moves most pinned <-> device copies with a blit KERNEL, which the copy trace does not list: they are the link's
        #  traffic all the same -- direction unknown from the trace, "copy" here)
        ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy" if "copyBuffer" in nm else "kernel", nm, int(r.get("Queue_Id", 0) or 0)))
    for r in load(dirname, "memory_copy_trace"):
        d = r.get("Direction", "")
        kind = "h2d" if "HOST_TO_DEVICE" in d else "d2h" if "DEVICE_TO_HOST" in d else "d2d"
        ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind, d, int(r.get("Bytes", r.get("Size", 0)) or 0)))
    ops.sort()
    # the calls: clusters of operations separated by more than 100 ms; the last cluster with a large download is the last call
    clusters, cur = [], []
    for o in ops:
        if cur and o[0] - max(x[1] for x in cur) > 100e6:
            clusters.append(cur)
            cur = []
        cur.append(o)
    if cur:
        clusters.append(cur)
    # (this rocprofv3's copy trace carries no sizes: a copy counts as a slice's when it lasts 50 us or more; sizes come from the call's own totals)
    calls = [c for c in clusters if sum(o[1] - o[0] for o in c if o[2] in ("d2h", "copy")) > 3e6]
    last = calls[-1]
    t0 = last[0][0]
    ms = lambda t: (t - t0) / 1e6  # noqa: E731
    end = max(o[1] for o in last)
    by = {k: union([(ms(o[0]), ms(o[1])) for o in last if o[2] == k]) for k in ("h2d", "d2h", "kernel", "copy")}
    busy = {k: sum(b - a for a, b in v) for k, v in by.items()}
    d2h_big = [o for o in last if o[2] == "d2h" and o[1] - o[0] >= 50e3]
    h2d_big = [o for o in last if o[2] == "h2d" and o[1] - o[0] >= 50e3]
    nbytes = {"h2d": (meta or {}).get("in_bytes", 0), "d2h": (meta or {}).get("out_bytes", 0)}
    out = {
        "what": "one fg_transcode_batch call (the last of four), from rocprofv3 --kernel-trace --memory-copy-trace",
        "span_ms": round(ms(end), 3),
        "busy_ms": {k: round(v, 3) for k, v in busy.items()},
        "both_directions_busy_ms": round(overlap(by["h2d"], by["d2h"]), 3),
        "first_download_starts_ms": round(ms(d2h_big[0][0]), 3) if d2h_big else None,
        "last_upload_ends_ms": round(ms(max(o[1] for o in h2d_big)), 3) if h2d_big else None,
        "download_alone_at_the_end_ms": round(ms(end) - ms(max(o[1] for o in h2d_big)), 3) if h2d_big else None,
        "bytes": nbytes,
        "rate_while_busy_GBps": {k: round(nbytes[k] / (busy[k] * 1e-3) / 1e9, 2) if busy[k] else None for k in ("h2d", "d2h")},
        "either_direction_busy_ms": round(sum(b - a for a, b in union([tuple(x) for x in by["h2d"] + by["d2h"]])), 3),
        "d2h_idle_stretches_ms": [[round(a1, 3), round(b0, 3)] for (a0, a1), (b0, b1) in zip([[0, 0]] + by["d2h"], by["d2h"] + [[ms(end), ms(end)]]) if b0 - a1 > 0.05],
        "link_busy_ms (sdma copies + blit kernels, union)": round(sum(b - a for a, b in union([tuple(x) for x in by["h2d"] + by["d2h"] + by["copy"]])), 3),
        "link_idle_stretches_ms": [[round(a1, 3), round(b0, 3)] for (a0, a1), (b0, b1) in
                                   zip([[0, 0]] + union([tuple(x) for x in by["h2d"] + by["d2h"] + by["copy"]]),
                                       union([tuple(x) for x in by["h2d"] + by["d2h"] + by["copy"]]) + [[ms(end), ms(end)]]) if b0 - a1 > 0.03],
        "ops (>= 20 us; start, end, what, queue)": [[round(ms(o[0]), 3), round(ms(o[1]), 3), o[2] + ":" + o[3][-28:], o[4]] for o in last if o[1] - o[0] >= 20e3],
        "kernels": {},
        "copies": [{"kind": o[2], "start_ms": round(ms(o[0]), 3), "end_ms": round(ms(o[1]), 3)} for o in last if o[2] != "kernel" and o[1] - o[0] >= 50e3],
    }
    for o in last:
        if o[2] in ("kernel", "copy"):
            k = out["kernels"].setdefault(o[3], {"calls": 0, "ms": 0.0})
            k["calls"] += 1
            k["ms"] = round(k["ms"] + (o[1] - o[0]) / 1e6, 3)
    if meta:
        out["call"] = meta
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "report":
        meta = None
        if len(sys.argv) > 3:
            for ln in open(sys.argv[3]):
                if ln.startswith("{"):
                    meta = json.loads(ln)
        report(sys.argv[2], meta)
    else:
        run()
