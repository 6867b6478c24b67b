#!/usr/bin/env python3
"""fg_frame_decode_batch from a PINNED raw stream, launch-option variants alternated on one box: tools/probe/fused_host.py [workload ...]
  FG_PROBE_OPTS='k=v,k=v;...'   variants (the empty one = the library's own choices); no_fused_framing=1 = the form of rounds 3-5
  FG_PROBE_LINES                lines per call (default 1 000 000, as the driver's e2e legs)
Every variant: 1 warm-up + 5 calls, two rounds, best and median wall-clock per call; fg_decode_batch (zero-copy, framed lines) beside
them as the ceiling of what a kernel that reads the link itself reaches on this box."""
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch  # noqa: E402,F401

from flowgger_amd import GelfDecoder, LTSVDecoder, RFC5424Decoder, synth  # noqa: E402
from flowgger_amd import _lib as L  # noqa: E402


def gen(wl, n):
    if wl == "cfg3":
        return GelfDecoder(), synth.gelf_lines(n, invalid_frac=0.01)
    if wl == "ltsv":
        return LTSVDecoder(synth.LTSV_CONFIG), synth.ltsv_lines(n, invalid_frac=0.01)
    return RFC5424Decoder(), synth.rfc5424_lines(n, cfg=4 if wl == "cfg4" else 2, sd=wl == "cfg4", invalid_frac=0.01)


def pinned(nbytes):
    p = C.c_void_p()
    L.check(L.lib().fg_alloc_pinned(nbytes, C.byref(p)), "fg_alloc_pinned")
    return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (nbytes,)), p


def main():
    wls = sys.argv[1:] or ["cfg2", "cfg3", "cfg4"]
    n = int(os.environ.get("FG_PROBE_LINES", "1000000"))
    variants = os.environ.get("FG_PROBE_OPTS", ";no_fused_framing=1").split(";")
    lib = L.lib()
    lib.fg_set_pinned_limits(32 << 30, 1 << 30)
    for wl in wls:
        tile = min(n, 250_000)
        dec, lines = gen(wl, tile)
        reps = (n + tile - 1) // tile
        one = b"".join(ln + b"\n" for ln in lines)
        raw, hraw = pinned(len(one) * reps + 64)
        for r in range(reps):
            raw[r * len(one):(r + 1) * len(one)] = np.frombuffer(one, np.uint8)
        nbytes, nl = len(one) * reps, len(lines) * reps
        data, offsets = synth.pack(lines)
        tb = int(offsets[-1])
        pdata, hd = pinned(tb * reps + 64)
        poffs8, ho = pinned((nl + 1) * 8 + 64)
        poffs = poffs8[: (nl + 1) * 8].view(np.uint64)
        for r in range(reps):
            pdata[r * tb:(r + 1) * tb] = data[:tb]
            poffs[r * len(lines):(r + 1) * len(lines)] = offsets[:-1] + np.uint64(r * tb)
        poffs[nl] = tb * reps
        gb = (C.c_double * 3)()
        lib.fg_measure_link(dec._ctx, 1 << 30, gb)
        row = {"workload": wl, "lines": nl, "stream_bytes": nbytes, "link_GBps": [round(float(x), 1) for x in gb]}
        st = L.fg_tables()

        def timed(call):
            call()
            ts = []
            for _ in range(5):
                t0 = time.perf_counter()
                call()
                ts.append(time.perf_counter() - t0)
            return ts

        ts = timed(lambda: L.check(lib.fg_decode_batch(dec._ctx, dec.fmt, pdata.ctypes.data, tb * reps, poffs.ctypes.data, nl, C.byref(st)), "fg_decode_batch"))
        row["decode_batch_zero_copy"] = {"M_lines_s": round(nl / min(ts) / 1e6, 1), "of_link": round((tb * reps + 8 * nl) / min(ts) / 1e9 / gb[0], 3),
                                         "wall_ms": round(min(ts) * 1e3, 3)}

        def kernel_ms(call):  # (HIP events around the launch on the ctx's stream: what of the call is the kernel)
            lib.fg_set_timing(dec._ctx, 1)
            call()
            ms = C.c_float()
            lib.fg_last_kernel_ms(dec._ctx, C.byref(ms))
            lib.fg_set_timing(dec._ctx, 0)
            return round(float(ms.value), 3)

        row["decode_batch_zero_copy"]["kernel_ms"] = kernel_ms(lambda: L.check(lib.fg_decode_batch(dec._ctx, dec.fmt, pdata.ctypes.data, tb * reps, poffs.ctypes.data, nl, C.byref(st)), "fg_decode_batch"))
        for v in [x for x in os.environ.get("FG_PROBE_DB_OPTS", "").split(";") if x]:  # (the zero-copy launch under launch options of its own)
            dec.set_launch_opts(**{k: int(x) for k, x in (kv.split("=") for kv in v.split(",") if kv)})
            ts = timed(lambda: L.check(lib.fg_decode_batch(dec._ctx, dec.fmt, pdata.ctypes.data, tb * reps, poffs.ctypes.data, nl, C.byref(st)), "fg_decode_batch"))
            row["decode_batch_zero_copy:" + v] = {"M_lines_s": round(nl / min(ts) / 1e6, 1), "of_link": round((tb * reps + 8 * nl) / min(ts) / 1e9 / gb[0], 3), "wall_ms": round(min(ts) * 1e3, 3)}
        dec.set_launch_opts()
        res = {v: [] for v in variants}
        paths, kms = {}, {}
        for _round in range(2):
            for v in variants:
                opts = {k: int(x) for k, x in (kv.split("=") for kv in v.split(",") if kv)}
                dec.set_launch_opts(**opts)
                st2, po, nf, cons = L.fg_tables(), C.c_void_p(), C.c_uint64(), C.c_uint64()

                def call():
                    L.check(lib.fg_frame_decode_batch(dec._ctx, dec.fmt, L.FG_FRAME_LINE, raw.ctypes.data, nbytes, 1, C.byref(st2), C.byref(po), C.byref(nf),
                                                      C.byref(cons)), "fg_frame_decode_batch")
                    assert nf.value == nl, (nf.value, nl)

                res[v] += timed(call)
                paths[v] = int(lib.fg_last_host_path(dec._ctx))
                kms[v] = kernel_ms(call)
        dec.set_launch_opts()
        for v in variants:
            ts = sorted(res[v])
            row["frame_decode_batch" + (":" + v if v else "")] = {"best_M_lines_s": round(nl / ts[0] / 1e6, 1), "median_M_lines_s": round(nl / ts[len(ts) // 2] / 1e6, 1),
                                                                 "best_of_link": round(nbytes / ts[0] / 1e9 / gb[0], 3), "path": paths[v],
                                                                 "wall_ms": round(ts[0] * 1e3, 3), "kernel_ms": kms[v]}
        print(json.dumps(row), flush=True)
        for h in (hraw, hd, ho):
            lib.fg_free_pinned(h)


if __name__ == "__main__":
    main()
