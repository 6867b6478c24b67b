#!/usr/bin/env python3
"""Framing + decode of a raw stream resident in HBM, two ways, on one box: tools/probe/fused_frame.py [workload ...]
  two-step  fg_frame_device (one-pass scan; the frame count visits the host) + fg_decode_frames_device   (rounds 1-5)
  fused     fg_frame_decode_device: the decode kernel frames its tiles itself, one read of the stream   (round 6)
A tile of FG_PROBE_TILE lines (default 250 000) replicated FG_PROBE_REPS times (default 16), HIP events on the launch stream, median of
7 after 2 warm-ups.  FG_PROBE_OPTS='k=v,k=v;...' adds launch-option variants of the fused launch; FG_PROBE_AVG: the average line
length the fused launch is planned for (default: the corpus's own)."""
import json
import os
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch  # noqa: E402

from flowgger_amd import GelfDecoder, LTSVDecoder, RFC5424Decoder, synth  # noqa: E402
from flowgger_amd import _lib as L  # noqa: E402
from flowgger_amd.tables import DeviceTables  # noqa: E402


def gen(wl, n):
    if wl == "cfg3":
        return GelfDecoder(), synth.gelf_lines(n, invalid_frac=0.01)
    if wl == "ltsv":
        return LTSVDecoder(synth.LTSV_CONFIG), synth.ltsv_lines(n, invalid_frac=0.01)
    return RFC5424Decoder(), synth.rfc5424_lines(n, cfg=4 if wl == "cfg4" else 2, sd=wl == "cfg4", invalid_frac=0.01)


def timed(fn, stream, dev, reps=7, warm=2):
    for _ in range(warm):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(stream)
        fn()
        b.record(stream)
    torch.cuda.synchronize(dev)
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2], ts[0]


def main():
    wls = sys.argv[1:] or ["cfg2", "cfg4", "ltsv", "cfg3"]
    tile = int(os.environ.get("FG_PROBE_TILE", "250000"))
    reps = int(os.environ.get("FG_PROBE_REPS", "16"))
    variants = [v for v in os.environ.get("FG_PROBE_OPTS", "").split(";")]
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream(dev)
    for wl in wls:
        dec, lines = gen(wl, tile)
        raw = b"".join(ln + b"\n" for ln in lines)
        n = len(lines) * reps
        host = np.frombuffer(raw, np.uint8)
        d_bytes = torch.cat([torch.from_numpy(host.copy()).to(dev).repeat(reps), torch.zeros(64, dtype=torch.uint8, device=dev)])[: len(raw) * reps]
        nbytes = len(raw) * reps
        avg = int(os.environ.get("FG_PROBE_AVG", "0")) or (len(raw) + len(lines) - 1) // len(lines)
        tables = DeviceTables(n + 64, nbytes // 8 + 4096, dev)
        row = {"workload": wl, "lines": n, "bytes": nbytes, "avg_line": avg}
        # ---- two-step ----
        d_off, d_bad, nf = dec.frame_device(d_bytes, L.FG_FRAME_LINE, cap_frames=n + 16)
        assert nf == n
        t_frame = timed(lambda: dec.frame_device(d_bytes, L.FG_FRAME_LINE, cap_frames=n + 16), stream, dev)
        t_dec = timed(lambda: dec.decode_frames_device(d_bytes, d_off, n, tables, L.FG_FRAME_LINE, d_bad, stream), stream, dev)
        want_meta = tables.column("meta").view(torch.int32)[:n].clone()
        want_cnt = tables.column("ent_count").view(torch.int32)[:n].clone()
        row["two_step"] = {"frame_ms": round(t_frame[0], 3), "decode_ms": round(t_dec[0], 3), "sum_ms": round(t_frame[0] + t_dec[0], 3)}
        # ---- fused ----
        for v in variants:
            opts = {k: int(x) for k, x in (kv.split("=") for kv in v.split(",") if kv)}
            dec.set_launch_opts(**opts)
            tables.buf.zero_()
            f_off, f_res = dec.frame_decode_device(d_bytes, L.FG_FRAME_LINE, tables, n + 16, final=True, avg_line=avg)
            torch.cuda.synchronize(dev)
            res = f_res.cpu().numpy()
            ok = int(res[0]) == n and int(res[1]) == 0 and bool((f_off[: n + 1] == d_off[: n + 1]).all())
            ok = ok and bool((tables.column("meta").view(torch.int32)[:n] == want_meta).all()) and bool((tables.column("ent_count").view(torch.int32)[:n] == want_cnt).all())
            if not ok:
                gm, gc = tables.column("meta").view(torch.int32)[:n], tables.column("ent_count").view(torch.int32)[:n]
                wm, wc = want_meta, want_cnt
                dm, dc = (gm != wm).nonzero().flatten(), (gc != wc).nonzero().flatten()
                row["diff" + (":" + v if v else "")] = {"res": res.tolist(), "meta_diffs": int(dm.numel()), "count_diffs": int(dc.numel()),
                                                        "first_meta": [int(dm[0]), hex(int(gm[dm[0]]) & 0xFFFFFFFF), hex(int(wm[dm[0]]) & 0xFFFFFFFF)] if dm.numel() else None,
                                                        "first_count": [int(dc[0]), int(gc[dc[0]]), int(wc[dc[0]])] if dc.numel() else None,
                                                        "offsets_equal": bool((f_off[: n + 1] == d_off[: n + 1]).all())}
            t = timed(lambda: dec.frame_decode_device(d_bytes, L.FG_FRAME_LINE, tables, n + 16, final=True, avg_line=avg), stream, dev)
            if hasattr(L.lib(), "fg_debug_fused_stats"):  # (the measurement variant: FLOWGGER_AMD_LIB=libfg_hip_stats.so)
                import ctypes as C
                st = (C.c_uint64 * 16)()
                dec.frame_decode_device(d_bytes, L.FG_FRAME_LINE, tables, n + 16, final=True, avg_line=avg)
                L.lib().fg_debug_fused_stats(dec._ctx, st)
                tiles = max(int(st[2]), 1)
                names = ["frames", "abort", "tiles", "slow_lookbacks", "staged_on", "tail_scans", "passes", "err_tiles", "cyc_wait_window", "cyc_stage_a",
                         "cyc_count_tail", "cyc_prefetch_publish", "cyc_lookback_rows", "cyc_list_stage_b", "slow_tcount", "slow_blocks"]
                row["stats" + (":" + v if v else "")] = {nm: (int(st[i]) if (i < 8 or i >= 14) else round(int(st[i]) / tiles)) for i, nm in enumerate(names)}
            row["fused" + (":" + v if v else "")] = {"ms": round(t[0], 3), "best_ms": round(t[1], 3), "same_result": ok,
                                                     "G_lines_s": round(n / t[0] / 1e6, 3), "stream_TBps": round(nbytes / t[0] / 1e9, 3)}
        dec.set_launch_opts()
        print(json.dumps(row), flush=True)
        del d_bytes, tables


if __name__ == "__main__":
    main()
