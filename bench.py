#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: RFC5424 log lines/s on N x MI355X + achieved HBM GB/s.

Workload (N=1 and per GPU at N>1, weak scaling): BASELINE config[1] -- RFC5424 without structured
data, 100 M lines @ ~256 B (a deterministic 1 M-line tile, seed 0x54240002, replicated x100 in
HBM with rebased offsets, 1 % invalid lines).  A "step" = ONE pass of the hot path
(fg_decode_batch_device through the C ABI) over the whole resident batch: every line tokenised,
every table row written.  Inputs are resident in HBM when the timed region starts.

One JSON line on stdout (rank 0).  `roofline.achieved` = algorithmic bytes per launch (SURVEY 8d:
line bytes + 4 B offset read, 64 B table row written, + 20 B per SD entry) / mean kernel time
measured with HIP events on the launch stream.  `cpu_baseline` = the C++ oracle (a restatement of
the reference's CPU decoders, "port") timed on this box's host cores on a bounded sample, one
thread and all cores.  `e2e` = the PCIe-inclusive host-buffer entry points (never `value`).

`python bench.py --gpus N` launches its own N ranks (torch.distributed.run, one per GPU, RCCL for the
barrier / max-over-ranks) when no launcher has set WORLD_SIZE; under torchrun it uses the env as given.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--tile-lines", type=int, default=1_000_000)
    ap.add_argument("--reps", type=int, default=100, help="tile replicas resident per GPU")
    ap.add_argument("--workload", default="cfg2", choices=["cfg2", "cfg3", "cfg4", "cfg5", "ltsv", "ltsv5", "frame", "cfg1", "rfc3164"],
                    help="cfg2 = the BASELINE metric's configuration (default); the others time the remaining "
                         "kernels on their parity-test corpora (not bench lines, see DESIGN.md)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the PCIe-inclusive host-buffer legs (fg_decode_batch / fg_transcode_batch)")
    ap.add_argument("--spawn", action="store_true",
                    help="launch the ranks through torch.distributed.run even for --gpus 1 (the path --gpus N>1 takes by itself "
                         "when WORLD_SIZE is not set)")
    ap.add_argument("--line-len", type=int, nargs=2, default=None, metavar=("LO", "HI"),
                    help="cfg2 only: uniform line-length range instead of 192..320 (tuning experiments)")
    ap.add_argument("--invalid-frac", type=float, default=0.01, help="share of invalid lines in the tile (SURVEY 8d: 1 %%)")
    return ap.parse_args()


WORKLOADS = {
    # name: (format id, description)
    "cfg2": (0, "BASELINE configs[1]: RFC5424 no structured data"),
    "cfg4": (0, "BASELINE configs[3] shape: RFC5424 with structured data (~12 pairs)"),
    "cfg5": (0, "BASELINE configs[4] shape: RFC5424, log-uniform 64 B..8 KiB lines with structured data"),
    "cfg3": (2, "BASELINE configs[2]: GELF/JSON, 8 flat extra fields"),
    "ltsv": (1, "LTSV, typed schema (the LTSV half of BASELINE configs[4])"),
    "ltsv5": (1, "BASELINE configs[4] shape, LTSV half: log-uniform 64 B..8 KiB lines"),
    "frame": (0, "GPU framing + UTF-8 validation of the newline-terminated cfg2 stream (SURVEY 8f-1), then decode of the frames"),
    "rfc3164": (3, "RFC3164 (BSD syslog) decoder, both forms, 15 % with IANA zone names (SURVEY 8f-3)"),
    "cfg1": (0, "BASELINE configs[0] pipeline on the GPU: RFC5424 decode -> GELF encoder -> line merger (SURVEY 8f-2/8f-4), cfg2 corpus"),
}


def self_launch(args) -> int:
    """`python bench.py --gpus N` with no torchrun around it: re-exec under torch.distributed.run, one rank per GPU
    (the driver's own launch line for N>1 does the same and never gets here: it sets WORLD_SIZE)."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    argv = [a for a in sys.argv[1:] if a != "--spawn"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve()), *argv]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), FG_BENCH_SPAWNED="1")
    return subprocess.call(cmd, env=env)


def source_hash(workload=None) -> str:
    """Identity of the kernel sources a PMC traffic figure belongs to (profiles/traffic.json, tools/update_traffic.py): the files
    the workload's kernel object was compiled from."""
    from flowgger_amd.build import source_hash as h

    return h(workload)


def e2e_legs(dec, fmt, data, offsets, n_tile, tile_bytes, want_transcode):
    """PCIe-inclusive rates of the host-buffer entry points on a bounded sample (4 tiles from pinned memory):
    fg_decode_batch = H2D + kernels + D2H of the tables; fg_transcode_batch = H2D + decode + GELF encode + line merger +
    D2H of the encoded stream.  Reported beside `value`, never as `value`."""
    import ctypes as C

    from flowgger_amd import GelfEncoder
    from flowgger_amd import _lib as L

    reps = 4
    n = n_tile * reps

    def pinned(nbytes, dt):
        p = C.c_void_p()
        L.check(L.lib().fg_alloc_pinned(nbytes, C.byref(p)), "fg_alloc_pinned")
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (nbytes,)).view(dt), p

    pdata, hd = pinned(tile_bytes * reps + 32, np.uint8)
    poffs, ho = pinned((n + 1) * 8, np.uint64)
    for r in range(reps):
        pdata[r * tile_bytes:(r + 1) * tile_bytes] = data[:tile_bytes]
        poffs[r * n_tile:(r + 1) * n_tile] = offsets[:-1] + np.uint64(r * tile_bytes)
    poffs[n] = tile_bytes * reps
    out = {"sample": f"{n} lines ({reps} tiles) from pinned host memory, best of 3"}
    try:
        best = 1e9
        for _ in range(4):
            st = L.fg_tables()
            t0 = time.perf_counter()
            L.check(L.lib().fg_decode_batch(dec._ctx, fmt, pdata.ctypes.data, tile_bytes * reps, poffs.ctypes.data, n, C.byref(st)),
                    "fg_decode_batch")
            best = min(best, time.perf_counter() - t0)
        out["decode_batch"] = {"lines_per_s": n / best, "GBps_in": tile_bytes * reps / best / 1e9, "ms": best * 1e3,
                               "what": "fg_decode_batch: H2D + kernels + D2H of the tables"}
        if want_transcode:
            enc = GelfEncoder(None, merger="line")
            cfg, _keep = enc._cfg_struct(0.0)
            best, res = 1e9, L.fg_transcoded()
            for _ in range(4):
                t0 = time.perf_counter()
                L.check(L.lib().fg_transcode_batch(dec._ctx, fmt, L.FG_FRAME_NONE, C.byref(cfg), pdata.ctypes.data, tile_bytes * reps,
                                                   poffs.ctypes.data, n, 1, C.byref(res)), "fg_transcode_batch")
                best = min(best, time.perf_counter() - t0)
            out["transcode_batch"] = {"lines_per_s": n / best, "GBps_in_plus_out": (tile_bytes * reps + int(res.out_bytes)) / best / 1e9,
                                      "out_bytes": int(res.out_bytes), "ms": best * 1e3,
                                      "what": "fg_transcode_batch: H2D + decode + GELF encode + line merger + D2H of the stream"}
    finally:
        L.lib().fg_free_pinned(hd)
        L.lib().fg_free_pinned(ho)
    return out


def cpu_baseline(fmt, data, offsets, n_lines, cfg=None, pipeline=False):
    """Oracle timing leg (the ONLY place bench.py touches oracle/)."""
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle_binding

    o = oracle_binding.Oracle()
    if fmt == 3:
        from flowgger_amd import tzdb

        o.set_rfc3164(2026, tzdb.default_table())
    cores = os.cpu_count() or 1
    ENC_GELF, MERGE_LINE = oracle_binding.ENC_GELF, oracle_binding.MERGE_LINE

    def run(threads, budget_s):
        passes, secs, n_ok = 0, 0.0, 0
        t_end = time.time() + budget_s
        while passes < 1 or (time.time() < t_end and passes < 256):
            if pipeline:
                s, n_ok, _ = o.bench_pipeline(fmt, ENC_GELF, MERGE_LINE, data, offsets, threads, cfg)
            else:
                s, n_ok = o.bench(fmt, data, offsets, threads, cfg)
            secs += s
            passes += 1
        return n_lines * passes / secs, passes, n_ok

    one, p1, n_ok = run(1, 3.0)        # a few seconds of one core
    allc, pn, n_ok = run(cores, 4.0)   # a few seconds of wall time on every host core = tens of CPU-seconds
    what = "decode + GELF encode + line merger + null sink (SURVEY 8d configuration 1)" if pipeline else "decode, owned Record per line"
    return {
        "value": allc, "unit": "lines/s", "cores": cores, "kind": "port",
        "single_thread": {"value": one, "cores": 1, "passes": p1},
        "sample": f"{pn} passes over the {n_lines}-line tile of the same workload on {cores} threads (and {p1} on one thread): {what}; "
                  "C++ restatement of the Rust decoders/encoders, not the Rust build (unavailable here)",
    }, n_ok


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.spawn):
        raise SystemExit(self_launch(args))
    import torch
    import torch.distributed as dist

    from flowgger_amd import GelfDecoder, LTSVDecoder, RFC5424Decoder, synth
    from flowgger_amd.tables import DeviceTables

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the decode path has no CPU fallback")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    # ---- synthetic batch, resident in HBM -------------------------------------------------
    wl = args.workload
    fmt, wl_desc = WORKLOADS[wl]
    sd = wl in ("cfg4", "cfg5")
    if wl == "cfg3":
        lines = synth.gelf_lines(args.tile_lines, invalid_frac=args.invalid_frac)
    elif wl in ("ltsv", "ltsv5"):
        lines = synth.ltsv_lines(args.tile_lines, invalid_frac=args.invalid_frac, long_tail=wl == "ltsv5")
    elif wl == "rfc3164":
        lines = synth.rfc3164_lines(args.tile_lines, invalid_frac=args.invalid_frac)
    elif wl == "frame":
        lines = [ln + b"\n" for ln in synth.rfc5424_lines(args.tile_lines, cfg=2, invalid_frac=args.invalid_frac)]
    elif wl == "cfg5":
        lines = synth.rfc5424_lines(args.tile_lines, cfg=5, sd=True, invalid_frac=args.invalid_frac, long_tail=True)
    else:
        kw = {"lo": args.line_len[0], "hi": args.line_len[1]} if (args.line_len and not sd) else {}
        lines = synth.rfc5424_lines(args.tile_lines, cfg=4 if sd else 2, sd=sd, invalid_frac=args.invalid_frac, **kw)
    data, offsets = synth.pack(lines)
    n_tile, tile_bytes = len(lines), int(offsets[-1])
    del lines
    reps = args.reps
    n = n_tile * reps
    raw = torch.from_numpy(data[:tile_bytes]).to(dev)
    # (filled in place: `cat(raw.repeat(reps), pad)` holds the batch twice for a moment -- 110 GB for configs[3]'s share)
    d_bytes = torch.empty(tile_bytes * reps + 32, dtype=torch.uint8, device=dev)
    d_bytes[tile_bytes * reps:] = 0
    d_bytes[:tile_bytes * reps].view(reps, tile_bytes).copy_(raw.unsqueeze(0).expand(reps, tile_bytes))
    o = torch.from_numpy(offsets[:-1].astype(np.int64)).to(dev)
    base = torch.arange(reps, device=dev, dtype=torch.int64).repeat_interleave(n_tile) * tile_bytes
    d_offsets = torch.cat([o.repeat(reps) + base, torch.tensor([tile_bytes * reps], device=dev, dtype=torch.int64)])
    del base, o, raw
    # entry slots: one per 24 input bytes (the corpora hold one pair per 38-42 bytes; a batch that needs more reports
    # FG_ST_OVERFLOW rows and fails the Ok-count check below) -- 0.75 x the input in HBM instead of 2.25 x
    ent_cap = (tile_bytes * reps // 24 if wl != "cfg2" else 0) + 4096 + 64 * 1024 * 256
    tables = DeviceTables(n, ent_cap, dev)
    if fmt == 3:
        from flowgger_amd import RFC3164Decoder

        dec = RFC3164Decoder({"rfc3164": {"current_year": 2026}}, device=local)
    else:
        dec = (GelfDecoder(device=local) if fmt == 2 else LTSVDecoder(synth.LTSV_CONFIG, device=local) if fmt == 1
               else RFC5424Decoder(device=local))
    stream = torch.cuda.current_stream(dev)

    frame_ms = None
    if wl == "frame":
        from flowgger_amd import _lib as FL

        raw_stream = d_bytes[:tile_bytes * reps]
        f_off, f_bad, f_n = dec.frame_device(raw_stream, FL.FG_FRAME_LINE, cap_frames=n + 16)  # warm-up + result
        assert f_n == n and bool((f_off[:n + 1] == d_offsets).all()), "GPU framing disagrees with the generator's offsets"
        evf = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        evf[0].record(stream)
        for _ in range(3):
            dec.frame_device(raw_stream, FL.FG_FRAME_LINE, cap_frames=n + 16)
        evf[1].record(stream)
        torch.cuda.synchronize(dev)
        frame_ms = evf[0].elapsed_time(evf[1]) / 3

    encode_ms = []
    if wl == "cfg1":
        from flowgger_amd import GelfEncoder

        enc = GelfEncoder(merger="line")
        dec.decode_device(d_bytes, d_offsets, tables, stream)
        e_out, e_off = enc.encode_device(dec, d_bytes, d_offsets, n, tables, stream=stream)  # sizes the output buffer
        enc_bytes = int(e_out.numel())
        e_buf = torch.empty(enc_bytes + 4096, dtype=torch.uint8, device=dev)
        del e_out, e_off

    def step():
        if wl == "cfg1":
            dec.decode_device(d_bytes, d_offsets, tables, stream)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            enc.encode_device(dec, d_bytes, d_offsets, n, tables, stream=stream, out=e_buf)
            b.record(stream)
            encode_ms.append((a, b))
        elif wl == "frame":
            dec.decode_frames_device(raw_stream, f_off, n, tables, FL.FG_FRAME_LINE, f_bad, stream)
        else:
            dec.decode_device(d_bytes, d_offsets, tables, stream)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record(stream)
        step()
        b.record(stream)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    rank_ms = [kernel_ms]
    if world > 1:  # every rank's own kernel time (HIP events on its stream), gathered over RCCL
        g = [torch.zeros(1, device=dev, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(g, torch.tensor([kernel_ms], device=dev, dtype=torch.float64))
        rank_ms = [float(x.item()) for x in g]

    # ---- validity: every replica of the tile produced the SAME rows -- all fixed columns, not only the status
    #      (ent_first is the one column that legitimately differs: entry slices are placed by wave-level allocation) ----
    meta = tables.column("meta").view(torch.int32).view(reps, n_tile)
    status = meta & 0xFF
    n_ok_tile = int((status[0] == 0).sum().item())
    if not os.environ.get("FG_ABLATE"):
        for col, width in (("meta", 4), ("ts", 8), ("hostname", 8), ("appname", 8), ("procid", 8), ("msgid", 8), ("msg", 8),
                           ("full_msg", 8), ("ent_count", 4)):
            c = tables.column(col)[: n * width].view(torch.int64 if width == 8 else torch.int32).view(reps, n_tile)
            assert bool((c == c[0:1]).all()), f"replicas disagree in column {col}: work was skipped or corrupted"
    reserved = int(tables.column("ent_used").view(torch.int64)[0].item())
    assert reserved <= ent_cap, "entry table overflow"
    # entries written (slots are reserved in per-wave chunks; `reserved` is a little more than what the lines own)
    used = int(tables.column("ent_count")[: n * 4].view(torch.int32).to(torch.int64).sum().item())

    if rank == 0:
        # SURVEY 8d's algorithmic bytes: line + one u32 offset read; a 64-byte row (+ 8 + 20 per structured-data pair)
        # written.  What this layout really moves is a little more (u64 offsets, 68-byte row, 18-byte entries): reported
        # beside it as `moved_bytes_per_launch`; `achieved` / `frac` use the SURVEY figure.
        alg_read = tile_bytes * reps + 4 * n
        alg_written = 64 * n + 20 * used
        moved = tile_bytes * reps + 8 * (n + 1) + 68 * n + 18 * used
        achieved = (alg_read + alg_written) / (kernel_ms * 1e-3) / 1e9
        out = {
            "metric": "log lines/sec (RFC5424, 256B avg) at 1/2/4/8 MI355X; achieved HBM GB/s",
            "value": n * world * args.steps / elapsed,
            "unit": "lines/s",
            "n_gpus": dist.get_world_size() if world > 1 else 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {
                "workload": wl_desc +
                            f", {n} lines/GPU @ {tile_bytes / n_tile:.0f} B avg ({n_tile}-line tile x{reps} resident in HBM), "
                            f"{args.invalid_frac * 100:g}% invalid lines",
                "lines_per_gpu": n, "bytes_per_gpu": tile_bytes * reps,
                "parallelism": f"lines sharded {world}-way, no data-path collective",
                "ok_lines_per_tile": n_ok_tile,
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                "kernel": ("fg::k_rfc5424", "fg::k_ltsv", "fg::k_gelf", "fg::k_rfc3164")[fmt], "kernel_ms": kernel_ms,
                "algorithmic_bytes_per_launch": alg_read + alg_written,
                "moved_bytes_per_launch": moved, "moved_frac": moved / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                "read_only_frac": alg_read / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
            },
            "ranks": {"n": len(rank_ms), "kernel_ms": rank_ms, "kernel_ms_min": min(rank_ms), "kernel_ms_max": max(rank_ms),
                      "launcher": "self (torch.distributed.run)" if os.environ.get("FG_BENCH_SPAWNED") else
                                  "external (WORLD_SIZE set)" if "WORLD_SIZE" in os.environ else "single process"},
        }
        if wl == "cfg1":
            ems = float(np.mean([a.elapsed_time(b) for a, b in encode_ms[-args.steps:]]))
            out["encode"] = {"ms": ems, "out_bytes": enc_bytes, "lines_per_s": n / (ems * 1e-3),
                             "GBps_in_plus_out": (tile_bytes * reps + 68 * n + enc_bytes) / (ems * 1e-3) / 1e9,
                             "note": "fg_encode_device (count + scan + write kernels incl. the host sync for the total); "
                                     "value / ms_per_step cover decode + encode; roofline.* is the decode kernel alone"}
            out["roofline"]["kernel_ms"] = kernel_ms - ems
            out["roofline"]["achieved"] = (alg_read + alg_written) / ((kernel_ms - ems) * 1e-3) / 1e9
            out["roofline"]["frac"] = out["roofline"]["achieved"] / HBM_PEAK_GBPS
        if frame_ms is not None:
            out["framing"] = {"ms": frame_ms, "GBps": tile_bytes * reps / (frame_ms * 1e-3) / 1e9,
                              "note": "fg_frame_device: scan + prefix + emit kernels incl. the host sync that returns the frame count"}
        # HBM traffic from the PMC passes (tools/prof.sh -> profiles/traffic.json): only when it was measured on THESE
        # kernel sources -- a figure from older code is not reported
        tr = ROOT / "profiles" / "traffic.json"
        if tr.exists():
            try:
                t = json.loads(tr.read_text()).get(args.workload)
                # (per workload: the files its kernel object was compiled from; where the dependency files are missing that
                #  falls back to the hash over every kernel source, which the entry carries as src_hash_all)
                if t and (t.get("src_hash") == source_hash(wl) or (t.get("src_hash_all") and t.get("src_hash_all") == source_hash())):
                    out["roofline"]["traffic"] = t["hbm_bytes_per_line"] * n
                    out["roofline"]["traffic_profile"] = t.get("profile")
                elif t:
                    out["roofline"]["traffic_note"] = "profiles/traffic.json holds a figure for older kernel sources: not reported"
            except Exception:
                pass
        if world == 1 and not args.no_e2e and wl in ("cfg2", "cfg1", "cfg3", "cfg4", "ltsv"):
            try:
                out["e2e"] = e2e_legs(dec, fmt, data, offsets, n_tile, tile_bytes, want_transcode=wl in ("cfg2", "cfg1"))
            except Exception as e:  # the PCIe legs never take the bench line down
                out["e2e"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            cb, n_ok_cpu = cpu_baseline(fmt, data, offsets, n_tile, synth.LTSV_CONFIG if fmt == 1 else None, pipeline=wl == "cfg1")
            assert n_ok_cpu == n_ok_tile, f"GPU Ok count {n_ok_tile} != oracle Ok count {n_ok_cpu}"
            out["cpu_baseline"] = cb
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
