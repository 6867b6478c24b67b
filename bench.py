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
thread and all cores (persistent threads: start-up is outside the timed region).
`e2e` = the PCIe-inclusive host-buffer entry points (never `value`), run on EVERY rank at the same
time from pinned buffers allocated on the GPU's NUMA node, with the measured link peak
(fg_measure_link) beside them: `e2e.aggregate` is the whole job's rate -- the regime the
north_star's "1 B lines/s on 8 GPUs" lives in.
`--workload cfg5mix` = BASELINE configs[4]: a 50/50 tagged RFC5424 + LTSV long-tail stream decoded
as two sub-batches and put back in arrival order by the host gather (fg_merge_tables; at N > 1 also
fg_gather_tables over the ranks' tables): `gather_ms`.

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
    ap.add_argument("--tile-lines", type=int, default=None, help="lines of the generated tile (default 1 M; cfg5mix 400 K)")
    ap.add_argument("--reps", type=int, default=None, help="tile replicas resident per GPU (default 100; cfg5mix 10)")
    ap.add_argument("--workload", default="cfg2",
                    choices=["cfg2", "cfg3", "cfg4", "cfg5", "cfg5mix", "ltsv", "ltsv5", "frame", "cfg1", "rfc3164"],
                    help="cfg2 = the BASELINE metric's configuration (default); the others time the remaining "
                         "kernels on their parity-test corpora (not bench lines, see DESIGN.md)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--encoder", default="gelf", choices=["gelf", "ltsv", "rfc5424", "rfc3164"],
                    help="cfg1 only: the encoder of the pipeline leg (BASELINE configs[0] is GELF; the others for the encoders' own A/Bs -- "
                         "the CPU baseline is GELF's: pass --no-cpu-baseline with them)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the PCIe-inclusive host-buffer legs (fg_decode_batch / fg_transcode_batch)")
    ap.add_argument("--no-mix", action="store_true",
                    help="default workload, one GPU, no launcher: skip the bounded BASELINE configs[4] leg (a cfg5mix sample run as a child "
                         "process; its rate and gather_ms ride along in the line as `configs4`)")
    ap.add_argument("--no-legs", action="store_true",
                    help="default workload, one GPU: skip the bounded BASELINE configs[2] (GELF) and configs[3] (structured data) legs "
                         "(4 M lines each, in this process; they ride along in the line as `configs2` / `configs3`)")
    ap.add_argument("--no-calib", action="store_true", help="skip the same-process copy / read calibration (roofline.copy_GBps, read_GBps)")
    ap.add_argument("--dry-run-backend", default=None, choices=["gloo"],
                    help="NO GPU: walk the N-rank control flow of this script -- process group, every barrier / max / all-gather of the "
                         "timed region, the validity and calibration exchanges, the PCIe legs' per-leg barriers, cfg5mix's merge and "
                         "rank gather (fg_merge_tables, shard.gather_distributed, fg_gather_tables on synthetic tables) -- over gloo on "
                         "the CPU with the decode stubbed out, so that a hang in a collective is found before the first 8-GPU run "
                         "(tests/test_shard_cpu.py runs it at world size 2).  Prints a line marked \"dry_run\": true; no throughput in it "
                         "means anything")
    ap.add_argument("--spawn", action="store_true",
                    help="launch the ranks through torch.distributed.run even for --gpus 1 (the path --gpus N>1 takes by itself "
                         "when WORLD_SIZE is not set)")
    ap.add_argument("--launcher", default="auto", choices=["auto", "torchrun", "threads"],
                    help="--gpus N > 1 without an external launcher: torchrun = re-exec under torch.distributed.run (one PROCESS per GPU, "
                         "RCCL); threads = N host threads of THIS process, one ctx per thread on its own device (no torch.distributed: "
                         "barriers and reductions are thread barriers -- the lines shard without a data-path collective, so nothing else is "
                         "needed); auto = torchrun, and threads when that launch fails (VERDICT r4 item 10)")
    ap.add_argument("--thread-devices", default=None,
                    help="threads launcher: the device of every thread, e.g. 0,0 to run two ranks on ONE GPU (tests on a 1-GPU box)")
    ap.add_argument("--line-len", type=int, nargs=2, default=None, metavar=("LO", "HI"),
                    help="cfg2 only: uniform line-length range instead of 192..320 (tuning experiments)")
    ap.add_argument("--invalid-frac", type=float, default=0.01, help="share of invalid lines in the tile (SURVEY 8d: 1 %%)")
    ap.add_argument("--launch-opts", default="", help="fg_set_launch_opts overrides, e.g. lines_per_group=32,waves_per_cu=6 (tuning)")
    a = ap.parse_args()
    if a.tile_lines is None:
        a.tile_lines = 400_000 if a.workload == "cfg5mix" else 1_000_000
    if a.reps is None:
        a.reps = 10 if a.workload == "cfg5mix" else 100
    return a


WORKLOADS = {
    # name: (format id, description)
    "cfg2": (0, "BASELINE configs[1]: RFC5424 no structured data"),
    "cfg4": (0, "BASELINE configs[3] shape: RFC5424 with structured data (~12 pairs)"),
    "cfg5": (0, "BASELINE configs[4] shape: RFC5424, log-uniform 64 B..8 KiB lines with structured data"),
    "cfg5mix": (0, "BASELINE configs[4]: 50/50 tagged RFC5424 + LTSV stream, log-uniform 64 B..8 KiB lines, two sub-batches, host-side "
                   "ordered gather by arrival index"),
    "cfg3": (2, "BASELINE configs[2]: GELF/JSON, 8 flat extra fields"),
    "ltsv": (1, "LTSV, typed schema (the LTSV half of BASELINE configs[4])"),
    "ltsv5": (1, "BASELINE configs[4] shape, LTSV half: log-uniform 64 B..8 KiB lines"),
    "frame": (0, "GPU framing + UTF-8 validation of the newline-terminated cfg2 stream (SURVEY 8f-1), then decode of the frames"),
    "rfc3164": (3, "RFC3164 (BSD syslog) decoder, both forms, 15 % with IANA zone names (SURVEY 8f-3)"),
    "cfg1": (0, "BASELINE configs[0] pipeline on the GPU: RFC5424 decode -> GELF encoder -> line merger (SURVEY 8f-2/8f-4), cfg2 corpus"),
}
KERNELS = ("fg::k_rfc5424", "fg::k_ltsv", "fg::k_gelf", "fg::k_rfc3164")


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


def bind_to_gpu_numa_node(local: int, tid: int = 0):
    """One process per GPU: run this rank (and first-touch its pinned staging buffers) on the CPUs of the NUMA node the GPU hangs
    off, so that H2D / D2H do not cross the socket interconnect.  tid != 0: the THREAD with that kernel id (the threads launcher: one
    rank per host thread -- Linux affinities are per thread, so every rank's thread sits on its own GPU's node just as a rank's process
    does: VERDICT r5 item 10).  Best effort; returns what was done for the JSON line."""
    try:
        import torch

        p = torch.cuda.get_device_properties(local)
        bdf = f"{getattr(p, 'pci_domain_id', 0):04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        node = int(Path(f"/sys/bus/pci/devices/{bdf}/numa_node").read_text())
        if node < 0:
            return {"pci": bdf, "numa_node": None}
        cpus = set()
        for part in Path(f"/sys/devices/system/node/node{node}/cpulist").read_text().strip().split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        cpus &= os.sched_getaffinity(tid)
        if cpus:
            os.sched_setaffinity(tid, cpus)
        return {"pci": bdf, "numa_node": node, "cpus": len(cpus), "bound": "thread" if tid else "process"}
    except Exception as e:  # noqa: BLE001
        return {"numa_node": None, "note": repr(e)[:80]}


class Dist:
    """torch.distributed over RCCL when a launcher set WORLD_SIZE (also for one rank: the same code runs at N = 1 and N = 8).
    backend="gloo": the --dry-run-backend walk of the same collectives on the CPU (dev = cpu, nothing to synchronise)."""

    def __init__(self, dev, backend="nccl"):
        import torch
        import torch.distributed as dist

        self.torch, self.dist, self.dev = torch, dist, dev
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.on = "WORLD_SIZE" in os.environ
        self.gpu = backend == "nccl"
        if self.on:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            if self.gpu:
                dist.init_process_group("nccl", device_id=dev)
            else:
                dist.init_process_group(backend)

    def barrier(self):
        if self.on:
            self.dist.barrier()
        if self.gpu:
            self.torch.cuda.synchronize(self.dev)

    def max(self, v: float) -> float:
        if not self.on:
            return v
        t = self.torch.tensor([v], device=self.dev, dtype=self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def all(self, vals):
        """every rank's list of floats -> list (by rank) of lists"""
        if not self.on:
            return [list(vals)]
        mine = self.torch.tensor(list(vals), device=self.dev, dtype=self.torch.float64)
        g = [self.torch.zeros_like(mine) for _ in range(self.world)]
        self.dist.all_gather(g, mine)
        return [[float(x) for x in t.tolist()] for t in g]

    def close(self):
        if self.on:
            self.dist.barrier()
            self.dist.destroy_process_group()


class ThreadDist:
    """--launcher threads: the same interface as Dist for N host threads of one process (one rank per thread, each on its own device
    with its own ctx -- the concurrency contract fg_clone is tested for, tcp_input.rs:39-47: one decoder clone per connection thread)."""

    def __init__(self, dev, world, rank, shared):
        import torch

        self.torch, self.dev, self.world, self.rank, self.sh = torch, dev, world, rank, shared
        self.on, self.gpu, self.threads = world > 1, True, True

    def barrier(self):
        self.torch.cuda.synchronize(self.dev)
        self.sh["barrier"].wait()

    def all(self, vals):
        self.sh["slots"][self.rank] = [float(v) for v in vals]
        self.sh["barrier"].wait()
        res = [list(v) for v in self.sh["slots"]]
        self.sh["barrier"].wait()
        return res

    def max(self, v: float) -> float:
        return max(r[0] for r in self.all([v]))

    def close(self):
        self.sh["barrier"].wait()


def run_threads(args) -> int:
    """N ranks as N threads of this process.  Rank 0 prints the JSON line; an exception in any rank aborts the barrier for all."""
    import threading

    devs = [int(d) for d in args.thread_devices.split(",")] if args.thread_devices else list(range(args.gpus))
    assert len(devs) == args.gpus, "--thread-devices must name one device per rank"
    shared = {"barrier": threading.Barrier(args.gpus), "slots": [None] * args.gpus}
    errors = []

    def rank_main(rank):
        try:
            main(args, thread_rank=(rank, devs[rank], shared))
        except BaseException as e:  # noqa: BLE001
            errors.append((rank, e))
            shared["barrier"].abort()

    ts = [threading.Thread(target=rank_main, args=(r,), name=f"rank{r}") for r in range(args.gpus)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for rank, e in errors:
        if not isinstance(e, threading.BrokenBarrierError):
            print(f"rank {rank}: {e!r}", file=sys.stderr)
    return 1 if errors else 0


def pinned(nbytes, dt):
    import ctypes as C

    from flowgger_amd import _lib as L

    # (the harness keeps whole multi-GB batches page-locked: beyond the allocator's cap -- 8 GiB by default -- it would hand out PAGEABLE
    #  memory, and the legs would measure the runtime's staged copies instead of the link)
    L.check(L.lib().fg_set_pinned_limits(1 << 40, 256 << 20), "fg_set_pinned_limits")

    p = C.c_void_p()
    L.check(L.lib().fg_alloc_pinned(nbytes, C.byref(p)), "fg_alloc_pinned")
    return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (nbytes,)).view(dt), p


def e2e_legs(D: Dist, dec, fmt, data, offsets, n_tile, tile_bytes, want_transcode, want_stream):
    """PCIe-inclusive rates of the host-buffer entry points on a bounded sample (4 tiles from pinned memory), on EVERY rank at
    the same time (barrier, then ITERS calls per rank, wall clock around them; aggregate = all ranks' lines / the slowest rank):
      fg_decode_batch        pinned buffers: zero-copy (the kernels read the lines over the link and write the pinned tables)
      fg_frame_decode_batch  H2D of the raw "\\n" stream ONLY, framing + UTF-8 + decode on the GPU, tables written into pinned memory
      fg_transcode_batch     H2D, decode, GELF encode, line merger, D2H of the encoded stream
    with the link's measured peak (fg_measure_link) beside them.  Reported beside `value`, never as `value`."""
    import ctypes as C

    from flowgger_amd import GelfEncoder
    from flowgger_amd import _lib as L

    reps, iters = 4, 3
    n = n_tile * reps
    lib = L.lib()
    pdata, hd = pinned(tile_bytes * reps + 32, np.uint8)
    poffs, ho = pinned((n + 1) * 8, np.uint64)
    for r in range(reps):
        pdata[r * tile_bytes:(r + 1) * tile_bytes] = data[:tile_bytes]
        poffs[r * n_tile:(r + 1) * n_tile] = offsets[:-1] + np.uint64(r * tile_bytes)
    poffs[n] = tile_bytes * reps
    out = {"sample": f"{n} lines ({reps} tiles) per rank from pinned host memory, {iters} calls per rank, all {D.world} rank(s) at once"}
    frees = [hd, ho]

    def leg(name, call, in_bytes, what, extra=None):
        # (every rank reaches the barrier and the gather of every leg, whatever happens in its own calls: a rank that fails a leg
        #  reports it -- the other ranks must not be left waiting in a collective)
        err, dt = None, float("nan")
        try:
            call()  # warm-up: staging buffers at size
        except Exception as e:  # noqa: BLE001
            err = repr(e)[:200]
        D.barrier()
        if err is None:
            try:
                t0 = time.perf_counter()
                for _ in range(iters):
                    call()
                dt = time.perf_counter() - t0
            except Exception as e:  # noqa: BLE001
                err = repr(e)[:200]
        per_rank = [x[0] for x in D.all([dt if err is None else -1.0])]
        good = [t for t in per_rank if t > 0]
        if err is not None or len(good) != len(per_rank):
            out[name] = {"error": err or "another rank failed this leg", "what": what}
            return False
        slowest = max(per_rank)
        ent = {"lines_per_s": n * iters / dt, "GBps_in": in_bytes * iters / dt / 1e9, "ms": dt / iters * 1e3, "what": what,
               "aggregate": {"lines_per_s": n * iters * D.world / slowest, "GBps_in": in_bytes * iters * D.world / slowest / 1e9,
                             "ranks": D.world, "per_rank_lines_per_s": [n * iters / t for t in per_rank]}}
        if extra:
            ent.update(extra())
        out[name] = ent
        return True

    try:
        gb = (C.c_double * 3)()
        if lib.fg_measure_link(dec._ctx, 1 << 30, gb) != 0:
            gb[0] = gb[1] = gb[2] = 0.0
        link = [float(gb[0]), float(gb[1]), float(gb[2])]
        allr = D.all(link)
        out["link_peak"] = {"h2d_GBps": link[0], "d2h_GBps": link[1], "bidir_GBps": link[2],
                            "what": "fg_measure_link: hipMemcpyAsync of a pinned 1 GiB buffer, best of 3 (one rank at a time is NOT "
                                    "enforced: every rank measures at once, as the legs below run)",
                            "per_rank_h2d_GBps": [r[0] for r in allr], "per_rank_d2h_GBps": [r[1] for r in allr]}
        st = L.fg_tables()
        if leg("decode_batch",
               lambda: L.check(lib.fg_decode_batch(dec._ctx, fmt, pdata.ctypes.data, tile_bytes * reps, poffs.ctypes.data, n, C.byref(st)),
                               "fg_decode_batch"),
               tile_bytes * reps + 8 * (n + 1), "fg_decode_batch from pinned buffers: zero-copy -- ONE launch, the kernels read bytes + offsets in place over the link and write the table columns into pinned host memory"):
            out["decode_batch"]["frac_of_link_h2d"] = out["decode_batch"]["GBps_in"] / link[0] if link[0] else None
        if want_stream and not bool((data[:tile_bytes] == 0x0A).any()):
            # the same lines as ONE raw newline-terminated stream: what LineSplitter reads off the socket
            ln = np.diff(offsets.astype(np.int64))
            stream_bytes = tile_bytes + n_tile
            ps, hs = pinned(stream_bytes * reps + 32, np.uint8)
            frees.append(hs)
            one = np.empty(stream_bytes, np.uint8)
            dst = (offsets[:-1].astype(np.int64) + np.arange(n_tile))
            one[:] = 0x0A
            # (scatter the tile's lines into their slots: each line is followed by one "\n")
            idx = np.repeat(dst - offsets[:-1].astype(np.int64), ln) + np.arange(tile_bytes)
            one[idx] = data[:tile_bytes]
            for r in range(reps):
                ps[r * stream_bytes:(r + 1) * stream_bytes] = one
            st2, po, nf, cons = L.fg_tables(), C.c_void_p(), C.c_uint64(), C.c_uint64()

            def call_stream():
                L.check(lib.fg_frame_decode_batch(dec._ctx, fmt, L.FG_FRAME_LINE, ps.ctypes.data, stream_bytes * reps, 1, C.byref(st2),
                                                  C.byref(po), C.byref(nf), C.byref(cons)), "fg_frame_decode_batch")
                assert nf.value == n, (nf.value, n)

            PATHS = {0: "none", 1: "decode zero-copy", 2: "decode sliced", 3: "frame + decode in ONE launch (the decode kernel frames the pinned chunk itself)",
                     4: "upload slices + framing scan + decode per slice", 5: "upload, frame, count on the host, decode"}
            if leg("frame_decode_batch", call_stream, stream_bytes * reps,
                   "fg_frame_decode_batch from a pinned raw stream: ONE launch since round 6 -- the decode kernel reads the chunk in place over the link, frames its tiles itself (UTF-8 check included) and writes rows, entries and frame offsets straight into pinned host memory",
                   extra=lambda: {"path": PATHS.get(int(lib.fg_last_host_path(dec._ctx)), "?")}):
                out["frame_decode_batch"]["frac_of_link_h2d"] = out["frame_decode_batch"]["GBps_in"] / link[0] if link[0] else None
            # the same leg the way rounds 3-5 ran it (upload slices, a framing scan, a capped decode grid per slice): same box, same buffers
            try:
                dec.set_launch_opts(**dict(getattr(dec, "_bench_opts", {}), no_fused_framing=True))
                if leg("frame_decode_batch_two_step", call_stream, stream_bytes * reps,
                       "the same call with FG_LO_NO_FUSED_FRAMING: H2D of the raw stream by sliced hipMemcpy, framing scan + decode per slice (rounds 3-5)",
                       extra=lambda: {"path": PATHS.get(int(lib.fg_last_host_path(dec._ctx)), "?")}):
                    out["frame_decode_batch_two_step"]["frac_of_link_h2d"] = out["frame_decode_batch_two_step"]["GBps_in"] / link[0] if link[0] else None
            finally:
                dec.set_launch_opts(**getattr(dec, "_bench_opts", {}))
        if want_transcode:
            enc = GelfEncoder(None, merger="line")
            cfg, _keep = enc._cfg_struct(0.0)
            res = L.fg_transcoded()
            if leg("transcode_batch",
                   lambda: L.check(lib.fg_transcode_batch(dec._ctx, fmt, L.FG_FRAME_NONE, C.byref(cfg), pdata.ctypes.data, tile_bytes * reps,
                                                          poffs.ctypes.data, n, 1, C.byref(res)), "fg_transcode_batch"),
                   tile_bytes * reps + 8 * (n + 1),
                   "fg_transcode_batch: H2D, decode, GELF encode, line merger, D2H of the stream (D2H-bound: the GELF text is 2.4x the input)",
                   extra=lambda: {"out_bytes": int(res.out_bytes)}):
                t = out["transcode_batch"]
                t["GBps_out"] = t["out_bytes"] / (t["ms"] * 1e-3) / 1e9
                t["frac_of_link_d2h"] = t["GBps_out"] / link[1] if link[1] else None
    finally:
        for h in frees:
            lib.fg_free_pinned(h)
    out["aggregate"] = {k: out[k]["aggregate"]["lines_per_s"] for k in ("decode_batch", "frame_decode_batch", "transcode_batch")
                        if k in out and "aggregate" in out[k]}
    return out


def cpu_baseline(legs, pipeline=False):
    """Oracle timing leg (the ONLY place bench.py touches oracle/).  legs = [(fmt, data, offsets, n_lines, cfg)] (one per
    sub-batch; their times add).  Persistent threads (oracle/fg_oracle.cpp fgo_bench_timed): created and parked before the
    clock starts, each walks the whole tile for the budget -- thread start-up and allocator warm-up are outside the figure."""
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle_binding

    o = oracle_binding.Oracle()
    if any(fmt == 3 for fmt, *_ in legs):
        from flowgger_amd import tzdb

        o.set_rfc3164(2026, tzdb.default_table())
    # the cores this process may really use: its affinity mask, cut down to the container's CPU quota (cgroup v2 cpu.max) --
    # 256 hardware threads are visible on the GPU boxes of this pool but the cgroup grants 16 CPUs: more threads than that only
    # add throttling (measured: linear to 16 threads, 130 M lines/s, then DOWN to 68 M at 256)
    visible = len(os.sched_getaffinity(0)) or os.cpu_count() or 1
    cores, quota = visible, None
    try:
        q, per = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if q != "max":
            quota = float(q) / float(per)
            cores = max(1, min(visible, int(quota)))
    except Exception:  # noqa: BLE001
        pass
    enc, mrg = (oracle_binding.ENC_GELF, oracle_binding.MERGE_LINE) if pipeline else (-1, 0)
    n_ok, total = 0, 0
    sec_one, sec_all = 0.0, 0.0  # seconds per line, summed over the sub-batches weighted by their share
    for fmt, data, offsets, n_lines, cfg in legs:
        _, ok = o.bench(fmt, data, offsets, 1, cfg)  # one plain pass: the Ok count the GPU must reproduce
        n_ok += ok
        s1, l1 = o.bench_timed(fmt, data, offsets, 1, 3.0 / len(legs), cfg, enc, mrg)
        sa, la = o.bench_timed(fmt, data, offsets, cores, 4.0 / len(legs), cfg, enc, mrg)
        sec_one += n_lines * s1 / l1
        sec_all += n_lines * sa / la
        total += n_lines
    one, allc = total / sec_one, total / sec_all
    what = "decode + GELF encode + line merger + null sink (SURVEY 8d configuration 1)" if pipeline else "decode, owned Record per line"
    return {
        "value": allc, "unit": "lines/s", "cores": cores, "kind": "port",
        "single_thread": {"value": one, "cores": 1},
        "parallel_efficiency": allc / (one * cores),
        "host_threads_visible": visible, "cgroup_cpu_quota": quota,
        "allocator": "glibc malloc, one arena per thread",
        "sample": f"~4 s of wall time on {cores} persistent threads (each walks the whole {total}-line tile of the same workload; thread start "
                  f"outside the timed region) and ~3 s on one thread: {what}; C++ restatement of the Rust decoders/encoders (-O3), not the "
                  "Rust build (unavailable here)",
    }, n_ok


def oracle_ok_count(fmt, data, offsets, cfg):
    """(checker, like cpu_baseline) the number of lines the oracle decodes to Ok"""
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle_binding

    _, ok = oracle_binding.Oracle().bench(fmt, data, offsets, 1, cfg)
    return ok


def cached_lines(key, gen):
    """FG_BENCH_CACHE=<dir> (measurement scripts that run this file many times on one box: tools/r05_final.sh): the generated tile is
    kept as one pickle per (workload, size, share of invalid lines) -- the generators are deterministic (fixed seeds), so this only
    saves the tens of seconds Python needs to format a million lines.  Unset: generate, as the driver's run does."""
    d = os.environ.get("FG_BENCH_CACHE")
    if not d:
        return gen()
    import pickle

    f = Path(d) / (key + ".pkl")
    if f.exists():
        return pickle.loads(f.read_bytes())
    lines = gen()
    f.parent.mkdir(parents=True, exist_ok=True)
    f.write_bytes(pickle.dumps(lines, protocol=4))
    return lines


def make_decoder(fmt, local, opts):
    from flowgger_amd import GelfDecoder, LTSVDecoder, RFC3164Decoder, RFC5424Decoder, synth

    if fmt == 3:
        dec = RFC3164Decoder({"rfc3164": {"current_year": 2026}}, device=local)
    else:
        dec = (GelfDecoder(device=local) if fmt == 2 else LTSVDecoder(synth.LTSV_CONFIG, device=local) if fmt == 1
               else RFC5424Decoder(device=local))
    if opts:
        dec.set_launch_opts(**opts)
    dec._bench_opts = dict(opts or {})
    return dec


class Resident:
    """One sub-batch resident in HBM: a tile replicated `reps` times with rebased offsets, its tables, its decoder."""

    def __init__(self, fmt, lines, reps, dev, local, opts, entries=True):
        import torch

        from flowgger_amd import synth
        from flowgger_amd.tables import DeviceTables

        self.fmt, self.reps = fmt, reps
        self.data, self.offsets = synth.pack(lines)
        self.n_tile, self.tile_bytes = len(lines), int(self.offsets[-1])
        self.n = self.n_tile * reps
        tb = self.tile_bytes
        raw = torch.from_numpy(self.data[:tb]).to(dev)
        # (filled in place: `cat(raw.repeat(reps), pad)` holds the batch twice for a moment -- 110 GB for configs[3]'s share)
        self.d_bytes = torch.empty(tb * reps + 32, dtype=torch.uint8, device=dev)
        self.d_bytes[tb * reps:] = 0
        self.d_bytes[:tb * reps].view(reps, tb).copy_(raw.unsqueeze(0).expand(reps, tb))
        o = torch.from_numpy(self.offsets[:-1].astype(np.int64)).to(dev)
        base = torch.arange(reps, device=dev, dtype=torch.int64).repeat_interleave(self.n_tile) * tb
        self.d_offsets = torch.cat([o.repeat(reps) + base, torch.tensor([tb * reps], device=dev, dtype=torch.int64)])
        del base, o, raw
        # entry slots: one per 24 input bytes (the corpora hold one pair per 38-42 bytes; a batch that needs more reports
        # FG_ST_OVERFLOW rows and fails the Ok-count check below) -- 0.75 x the input in HBM instead of 2.25 x
        self.ent_cap = (tb * reps // 24 if entries else 0) + 4096 + 64 * 1024 * 256
        self.tables = DeviceTables(self.n, self.ent_cap, dev)
        self.dec = make_decoder(fmt, local, opts)

    def decode(self, stream):
        self.dec.decode_device(self.d_bytes, self.d_offsets, self.tables, stream)

    def check_replicas(self):
        """every replica of the tile produced the SAME rows -- all fixed columns, not only the status (ent_first is the one
        column that legitimately differs: entry slices are placed by wave-level allocation); -> (Ok rows of one tile, entries)"""
        import torch

        t, reps, n_tile, n = self.tables, self.reps, self.n_tile, self.n
        meta = t.column("meta").view(torch.int32).view(reps, n_tile)
        n_ok_tile = int(((meta[0] & 0xFF) == 0).sum().item())
        if not os.environ.get("FG_ABLATE"):
            for col, width in (("meta", 4), ("ts", 8), ("hostname", 8), ("appname", 8), ("procid", 8), ("msgid", 8), ("msg", 8),
                               ("full_msg", 8), ("ent_count", 4)):
                c = t.column(col)[: n * width].view(torch.int64 if width == 8 else torch.int32).view(reps, n_tile)
                assert bool((c == c[0:1]).all()), f"replicas disagree in column {col}: work was skipped or corrupted"
        reserved = int(t.column("ent_used").view(torch.int64)[0].item())
        assert reserved <= self.ent_cap, "entry table overflow"
        # entries written (slots are reserved in per-wave chunks; `reserved` is a little more than what the lines own)
        used = int(t.column("ent_count")[: n * 4].view(torch.int32).to(torch.int64).sum().item())
        return n_ok_tile, used


def stamped(workload_key):
    """profiles/traffic.json's entry for a workload when it was measured on THESE kernel sources (tools/update_traffic.py stamps the PMC
    passes of the closing run with the hash of the files the workload's kernel object is compiled from), else None."""
    tr = ROOT / "profiles" / "traffic.json"
    try:
        t = json.loads(tr.read_text()).get(workload_key)
        if t and (t.get("src_hash") == source_hash(workload_key) or (t.get("src_hash_all") and t.get("src_hash_all") == source_hash())):
            return t
    except Exception:  # noqa: BLE001
        pass
    return None


def compute_roof(t):
    """The COMPUTE side of a kernel from its stamped instruction counters (`0.12 of HBM` says nothing for a kernel that
    sits at 0.85 of the VALU issue rate): wave-instructions per line and VALU busy = SQ_INSTS_VALU x 4 cycles / (kernel time x 2.4 GHz x
    1024 SIMDs) of the run the counters came from."""
    c = (t or {}).get("compute")
    if not c:
        return None
    return {"valu_per_line": c["valu_per_line"], "salu_per_line": c["salu_per_line"], "valu_busy": c["valu_busy"],
            "what": "rocprofv3 --pmc SQ_INSTS_VALU / SQ_INSTS_SALU of this kernel on these sources (" + str(t.get("profile")) + "): wave-instructions per "
                    "line, and VALU busy = VALU instructions x 4 cycles / (kernel time x 2.4 GHz x 1024 SIMDs) -- a wave64 VALU instruction "
                    "occupies its SIMD for 4 cycles; near 1.0 = bound by instruction issue, not by HBM"}


def bounded_leg(desc, fmt, lines, reps, dev, local, steps=10, warmup=2, D=None, stamp_key=None):
    """One more BASELINE configuration on a bounded resident sample (4 M lines), timed like the main workload (HIP events on the launch
    stream, replicas compared, Ok count returned for the oracle check): rides in the default line as configs2 / configs3 so that the
    driver's run observes every BASELINE configuration, not only configs[1]."""
    import torch

    R = Resident(fmt, lines, reps, dev, local, {}, entries=True)
    stream = torch.cuda.current_stream(dev)
    for _ in range(warmup):
        R.decode(stream)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for a, b in ev:
        a.record(stream)
        R.decode(stream)
        b.record(stream)
    torch.cuda.synchronize(dev)
    ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    n_ok_tile, used = R.check_replicas()
    alg_read = R.tile_bytes * reps + 4 * R.n
    alg_written = 64 * R.n + 20 * used
    out = {"workload": f"{desc}, {R.n} lines @ {R.tile_bytes / R.n_tile:.0f} B avg ({R.n_tile}-line tile x{reps} resident in HBM)",
           "value": R.n / (ms * 1e-3), "unit": "lines/s", "kernel": KERNELS[fmt], "kernel_ms": ms, "steps": steps,
           "roofline_frac": (alg_read + alg_written) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
           "read_only_frac": alg_read / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
           "achieved_GBps": (alg_read + alg_written) / (ms * 1e-3) / 1e9,
           "algorithmic_bytes_per_launch": alg_read + alg_written, "entries": used, "ok_lines_per_tile": n_ok_tile,
           "what": "bounded sample of this BASELINE configuration, same timing and replica checks as the main line; full size: "
                   "python bench.py --workload " + ("cfg3" if fmt == 2 else "cfg4 --reps 125")}
    if D is not None:  # the PCIe-inclusive entry points on this corpus too (pinned buffers: fg_decode_batch takes its zero-copy form)
        try:
            e = e2e_legs(D, R.dec, fmt, R.data, R.offsets, R.n_tile, R.tile_bytes, want_transcode=False, want_stream=True)
            out["e2e"] = {k: {kk: e[k][kk] for kk in ("lines_per_s", "GBps_in", "ms", "frac_of_link_h2d", "path", "what") if kk in e[k]}
                          for k in ("decode_batch", "frame_decode_batch", "frame_decode_batch_two_step") if k in e and "lines_per_s" in e[k]}
            out["e2e"]["sample"] = e.get("sample")
        except AssertionError:
            raise
        except Exception as ex:  # noqa: BLE001
            out["e2e"] = {"error": repr(ex)[:200]}
    cr = compute_roof(stamped(stamp_key)) if stamp_key else None
    if cr:
        out["compute"] = cr
    leg = (R.fmt, R.data, R.offsets, R.n_tile, None)
    del R
    torch.cuda.empty_cache()
    return out, leg


def small_batch_leg(dev, local, invalid_frac):
    """Kernel-only rate at the batch sizes a framer really hands over (the kernels were tuned at 4 M .. 100 M
    lines): 64 K / 256 K / 1 M lines of the headline corpus and of the long-tail structured-data corpus (64 B .. 8 KiB), resident in
    HBM, median of 9 launches after 3 warm-ups, HIP events on the launch stream.  One 64 K-line tile per corpus, replicated."""
    import torch

    from flowgger_amd import synth

    stream = torch.cuda.current_stream(dev)
    out = {"what": "fg_decode_batch_device on a resident batch of n lines (a 65 536-line tile x 1 / 4 / 16), kernel-only lines/s, median of 9"}
    for key, gen, entries in (("cfg2", lambda: synth.rfc5424_lines(65_536, cfg=2, invalid_frac=invalid_frac), False),
                              ("long_tail", lambda: synth.rfc5424_lines(65_536, cfg=5, sd=True, invalid_frac=invalid_frac, long_tail=True), True)):
        lines = gen()
        res = {}
        for reps in (1, 4, 16):
            R = Resident(0, lines, reps, dev, local, {}, entries=entries)
            for _ in range(3):
                R.decode(stream)
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(9)]
            for a, b in ev:
                a.record(stream)
                R.decode(stream)
                b.record(stream)
            torch.cuda.synchronize(dev)
            ms = sorted(a.elapsed_time(b) for a, b in ev)[4]
            R.check_replicas()
            res[str(R.n)] = {"lines_per_s": R.n / (ms * 1e-3), "kernel_us": ms * 1e3, "avg_line_bytes": R.tile_bytes / R.n_tile}
            del R
            torch.cuda.empty_cache()
        out[key] = res
    return out


def calibrate(dec, d_bytes, nbytes, dev, reps=3):
    """What THIS box's memory system gives a plain streaming kernel over the very buffer the decoder reads (fg_calibrate_device):
    a float4 copy (2 x nbytes of traffic) and a read-only sweep.  Three boxes of this pool differ by 8 % on the same code."""
    import torch

    from flowgger_amd import _lib as L

    lib = L.lib()
    stream = torch.cuda.current_stream(dev)
    nbytes = nbytes // 16 * 16
    dst = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    out = {}
    # (three spellings of the float4 copy -- grid-stride, non-temporal, one element per thread -- and the BEST is the box's copy rate:
    #  the ceiling the decoder is priced against must not be an artefact of one spelling)
    for name, mode, traffic in (("copy_gridstride_GBps", 0, 2 * nbytes), ("copy_nt_GBps", 2, 2 * nbytes), ("copy_flat_GBps", 3, 2 * nbytes),
                                ("read_GBps", 1, nbytes)):
        L.check(lib.fg_calibrate_device(dec._ctx, mode, d_bytes.data_ptr(), dst.data_ptr(), nbytes, stream.cuda_stream), "fg_calibrate_device")
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in ev:
            a.record(stream)
            L.check(lib.fg_calibrate_device(dec._ctx, mode, d_bytes.data_ptr(), dst.data_ptr(), nbytes, stream.cuda_stream), "fg_calibrate_device")
            b.record(stream)
        torch.cuda.synchronize(dev)
        out[name] = traffic / (min(a.elapsed_time(b) for a, b in ev) * 1e-3) / 1e9
    # (a figure above anything HBM3E can move is a launch that did not run, not a ceiling)
    sane = [out[k] for k in ("copy_gridstride_GBps", "copy_nt_GBps", "copy_flat_GBps") if out[k] <= 1.25 * HBM_PEAK_GBPS]
    if not sane:
        raise RuntimeError(f"no plausible copy figure: {out}")
    out["copy_GBps"] = max(sane)
    del dst
    torch.cuda.empty_cache()
    return out


def dry_run(args):
    """--dry-run-backend gloo: the collectives of main() in main()'s order, on the CPU, with the decode stubbed out (see parse_args).
    Kept beside main() on purpose: a collective added there belongs here too (tests/test_shard_cpu.py::test_bench_dry_run_n_ranks)."""
    import torch

    from flowgger_amd import shard

    D = Dist(torch.device("cpu"), backend=args.dry_run_backend)
    world, rank = D.world, D.rank
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    wl = args.workload
    for _ in range(args.warmup):
        pass
    D.barrier()                                    # before the timed region
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pass
    D.barrier()                                    # after it
    elapsed = D.max(time.perf_counter() - t0)      # MAX over ranks
    rank_ms = [r[0] for r in D.all([0.001 * (rank + 1)])]          # every rank's kernel time
    calib_all = D.all([5000.0 + rank, 6000.0 + rank])              # calibration exchange
    gather = None
    if wl == "cfg5mix":
        # two synthetic sub-batch tables per rank (one entry per second row), merged by arrival index, then every rank's table to rank 0
        n = 1000 + 10 * rank
        parts, index = [], []
        rng = np.random.default_rng(rank)
        tag = rng.integers(0, 2, n).astype(np.uint8)
        for k in (0, 1):
            m = int((tag == k).sum())
            t = shard._alloc_tables(m, max(m // 2, 1))
            t.a["meta"][:m] = np.arange(m, dtype=np.uint32) << 8
            t.a["ent_count"][:m] = (np.arange(m) % 2 == 0).astype(np.uint32)
            t.a["ent_first"][:m] = np.cumsum(np.concatenate([[0], t.a["ent_count"][:m - 1]])).astype(np.uint32) if m else 0
            t.a["ent_used"][0] = int(t.a["ent_count"][:m].sum())
            parts.append(t)
            index.append(np.flatnonzero(tag == k).astype(np.uint64))
        merged, src = shard.merge_tables(parts, index)
        assert merged.n == n and np.array_equal(src[:n], tag)
        D.barrier()
        full = shard.gather_distributed(merged, dst=0)
        D.barrier()
        g_ms = D.max(1.0)
        if rank == 0:
            assert full.n == sum(1000 + 10 * r for r in range(world))
        gather = {"ranks_gather_ms": g_ms, "all_ranks_gather_ms": [r[0] for r in D.all([g_ms])], "ranks_rows": int(full.n) if rank == 0 else None}
    e2e = None
    if not args.no_e2e and wl in ("cfg2", "cfg1", "cfg3", "cfg4", "ltsv"):
        e2e = {"link_peak": D.all([57.0, 57.0, 80.0])}
        for leg in ("decode_batch", "frame_decode_batch", "frame_decode_batch_two_step") + (("transcode_batch",) if wl in ("cfg2", "cfg1") else ()):
            D.barrier()                            # every rank starts a leg together ...
            e2e[leg] = [x[0] for x in D.all([0.01 * (rank + 1)])]   # ... and hands in its time
    if rank == 0:
        print(json.dumps({"dry_run": True, "backend": args.dry_run_backend, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "workload": wl, "elapsed_s": elapsed, "ranks": {"n": len(rank_ms), "kernel_ms": rank_ms},
                          "calibration_ranks": len(calib_all), "gather": gather, "e2e_legs": sorted(e2e) if e2e else None,
                          "what": "control flow only: process group, barriers, reductions and gathers of bench.py's N-rank run on the CPU; "
                                  "no decode ran and no number here is a measurement"}), flush=True)
    D.close()


def main(args=None, thread_rank=None):
    if args is None:
        args = parse_args()
    if thread_rank is None and "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.spawn) and not args.dry_run_backend:
        if args.launcher == "threads":
            raise SystemExit(run_threads(args))
        rc = self_launch(args)
        if rc != 0 and args.launcher == "auto" and args.gpus > 1:
            print(f"bench.py: torch.distributed.run exited with {rc}: running the {args.gpus} ranks as threads of one process", file=sys.stderr)
            rc = run_threads(args)
        raise SystemExit(rc)
    if thread_rank is None and "WORLD_SIZE" not in os.environ and args.dry_run_backend and (args.gpus > 1 or args.spawn):
        raise SystemExit(self_launch(args))
    if args.dry_run_backend:
        return dry_run(args)
    import torch

    from flowgger_amd import synth

    local = int(os.environ.get("LOCAL_RANK", "0")) if thread_rank is None else thread_rank[1]
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the decode path has no CPU fallback")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    all_cpus = os.sched_getaffinity(0)
    if thread_rank is None:
        numa = bind_to_gpu_numa_node(local)
        D = Dist(dev)
    else:  # (a rank = a host thread: ITS affinity goes to the GPU's node; pinned buffers are first-touched by the thread that fills them)
        import threading

        numa = bind_to_gpu_numa_node(local, tid=threading.get_native_id())
        D = ThreadDist(dev, args.gpus, thread_rank[0], thread_rank[2])
    world, rank = D.world, D.rank
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    opts = {k: int(v) for k, v in (kv.split("=") for kv in args.launch_opts.split(",") if kv)}

    # ---- synthetic batch, resident in HBM -------------------------------------------------
    wl = args.workload
    fmt, wl_desc = WORKLOADS[wl]
    sd = wl in ("cfg4", "cfg5")
    index = None
    ckey = f"{wl}_{args.tile_lines}_{args.invalid_frac:g}" + (f"_{args.line_len[0]}_{args.line_len[1]}" if args.line_len else "")
    if wl == "cfg5mix":
        tag, (la, ia), (lb, ib) = cached_lines(ckey, lambda: synth.mixed_cfg5(args.tile_lines, invalid_frac=args.invalid_frac))
        subs = [Resident(0, la, args.reps, dev, local, opts), Resident(1, lb, args.reps, dev, local, opts)]
        # arrival position of row j of replica r of a sub-batch: r * tile + position inside the tile
        index = [np.concatenate([ix + np.uint64(r * args.tile_lines) for r in range(args.reps)]) for ix in (ia, ib)]
        del la, lb
    else:
        def gen_lines():
            if wl == "cfg3":
                return synth.gelf_lines(args.tile_lines, invalid_frac=args.invalid_frac)
            if wl in ("ltsv", "ltsv5"):
                return synth.ltsv_lines(args.tile_lines, invalid_frac=args.invalid_frac, long_tail=wl == "ltsv5")
            if wl == "rfc3164":
                return synth.rfc3164_lines(args.tile_lines, invalid_frac=args.invalid_frac)
            if wl == "frame":
                return [ln + b"\n" for ln in synth.rfc5424_lines(args.tile_lines, cfg=2, invalid_frac=args.invalid_frac)]
            if wl == "cfg5":
                return synth.rfc5424_lines(args.tile_lines, cfg=5, sd=True, invalid_frac=args.invalid_frac, long_tail=True)
            kw = {"lo": args.line_len[0], "hi": args.line_len[1]} if (args.line_len and not sd) else {}
            return synth.rfc5424_lines(args.tile_lines, cfg=4 if sd else 2, sd=sd, invalid_frac=args.invalid_frac, **kw)

        lines = cached_lines(ckey if wl != "cfg1" else ckey.replace("cfg1", "cfg2"), gen_lines)
        subs = [Resident(fmt, lines, args.reps, dev, local, opts, entries=wl != "cfg2")]
        del lines
    R = subs[0]
    dec, tables, d_bytes, d_offsets = R.dec, R.tables, R.d_bytes, R.d_offsets
    n_tile, tile_bytes, reps = sum(s.n_tile for s in subs), sum(s.tile_bytes for s in subs), args.reps
    n = n_tile * reps
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
        # round 6: the decode kernel frames the stream ITSELF (fg_frame_decode_device): one read of the stream, nothing visits the host
        fused = None
        try:
            f_tab = R.tables
            fo, fr_ = dec.frame_decode_device(raw_stream, FL.FG_FRAME_LINE, f_tab, n, final=True, avg_line=(tile_bytes + n_tile - 1) // n_tile)
            torch.cuda.synchronize(dev)
            res = fr_.cpu().numpy()
            same = int(res[0]) == n and int(res[1]) == 0 and bool((fo[:n + 1] == d_offsets).all())
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
            for a, b in evs:
                a.record(stream)
                dec.frame_decode_device(raw_stream, FL.FG_FRAME_LINE, f_tab, n, final=True, avg_line=(tile_bytes + n_tile - 1) // n_tile)
                b.record(stream)
            torch.cuda.synchronize(dev)
            fms = sorted(a.elapsed_time(b) for a, b in evs)[2]
            fused = {"ms": fms, "stream_GBps": tile_bytes * reps / (fms * 1e-3) / 1e9, "lines_per_s": n / (fms * 1e-3), "frames_and_offsets_equal": same,
                     "what": "fg_frame_decode_device: ONE kernel frames (UTF-8 check included) and decodes the resident stream -- the stream is read "
                             "once and no count visits the host between framing and decode; beside it: framing.ms + the decode of the frames"}
            assert same, "the fused launch disagrees with the generator's offsets"
        except AssertionError:
            raise
        except Exception as e:  # noqa: BLE001
            fused = {"error": repr(e)[:200]}

    encode_ms = []
    if wl == "cfg1":
        import flowgger_amd as FA

        enc = {"gelf": FA.GelfEncoder, "ltsv": FA.LTSVEncoder, "rfc5424": FA.RFC5424Encoder, "rfc3164": FA.RFC3164Encoder}[args.encoder](merger="line")
        dec.decode_device(d_bytes, d_offsets, tables, stream)
        e_out, e_off = enc.encode_device(dec, d_bytes, d_offsets, n, tables, stream=stream)  # sizes the output buffer
        enc_bytes = int(e_out.numel())
        e_buf = torch.empty(enc_bytes + 4096, dtype=torch.uint8, device=dev)
        del e_out, e_off

    sub_ev = []  # cfg5mix: (start, between, end) per step -> the two kernels' times

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
        elif wl == "cfg5mix":
            mid = torch.cuda.Event(enable_timing=True)
            subs[0].decode(stream)
            mid.record(stream)
            subs[1].decode(stream)
            sub_ev.append(mid)
        else:
            R.decode(stream)

    for _ in range(args.warmup):
        step()
    D.barrier()
    sub_ev.clear()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record(stream)
        step()
        b.record(stream)
    D.barrier()
    elapsed = D.max(time.perf_counter() - t0)
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    rank_ms = [r[0] for r in D.all([kernel_ms])]  # every rank's own kernel time (HIP events on its stream), gathered over RCCL

    # ---- validity --------------------------------------------------------------------------
    checks = [s.check_replicas() for s in subs]
    n_ok_tile, used = sum(c[0] for c in checks), sum(c[1] for c in checks)

    # ---- same-process calibration: a float4 copy and a read-only sweep over the decoder's own resident buffer ----
    calib = None
    if not args.no_calib and wl != "cfg5mix":
        try:
            calib = calibrate(dec, d_bytes, tile_bytes * reps, dev)
        except Exception as e:  # noqa: BLE001 -- never takes the bench line down
            calib = {"error": repr(e)[:200]}
    calib_all = D.all([calib.get("copy_GBps", 0.0), calib.get("read_GBps", 0.0)]) if calib is not None else None

    # ---- configs[4]: the host-side ordered gather (SURVEY 8d "same + gather time") -----------
    gather = None
    if wl == "cfg5mix":
        from flowgger_amd import shard

        side = torch.cuda.Stream(dev)
        # round 4: the rows go back to their arrival positions ON THE DEVICE (fg_merge_tables_device: both sub-batches' tables are still in
        # HBM), and ONE merged table crosses the link.  (Rounds 2-3: D2H of both tables + fg_merge_tables on the host's cores -- 36 of 56 ms;
        # timed below as `host_merge_ms` for comparison.)
        d_index = [torch.from_numpy(np.ascontiguousarray(ix).astype(np.int64)).to(dev) for ix in index]
        d_merged, d_src = shard.merge_tables_device(subs[0].dec, [s.tables for s in subs], d_index)   # allocates the merged table
        merged = d_merged.to_host_pinned(side)  # (allocates + pins the staging buffer: a framer keeps it)
        torch.cuda.synchronize(dev)
        D.barrier()
        mev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        t0 = time.perf_counter()
        mev[0].record()
        d_merged, d_src = shard.merge_tables_device(subs[0].dec, [s.tables for s in subs], d_index, out=d_merged, d_src=d_src)
        mev[1].record()
        torch.cuda.synchronize(dev)
        merge_ms = (time.perf_counter() - t0) * 1e3
        merge_kernel_ms = mev[0].elapsed_time(mev[1])
        t0 = time.perf_counter()
        merged = d_merged.to_host_pinned(side, host=merged._pinned)  # D2H of the ONE merged table
        src = d_src.cpu().numpy()
        d2h_ms = (time.perf_counter() - t0) * 1e3
        assert merged.n == n and np.array_equal(src[: args.tile_lines], tag), "merge did not restore the arrival order"
        # the host merge of rounds 2-3, for comparison and as the check of the device merge (same rows at the same places)
        parts = [s.tables.to_host_pinned(side) for s in subs]
        hmerged, hsrc = shard.merge_tables(parts, index)
        t0 = time.perf_counter()
        hmerged, hsrc = shard.merge_tables(parts, index, out=hmerged, src=hsrc)
        host_merge_ms = (time.perf_counter() - t0) * 1e3
        # (round 5: the device merge compacts the entries -- dense, arrival order --, the host merge keeps the parts' layouts)
        assert np.array_equal(hsrc, src) and np.array_equal(hmerged.a["meta"][:n], merged.a["meta"][:n])
        assert np.array_equal(hmerged.a["ent_count"][:n], merged.a["ent_count"][:n])
        assert merged.ent_used == int(merged.a["ent_count"][:n].astype(np.int64).sum()) <= hmerged.ent_used
        # spot check: rows went back where they came from
        for k, s_ in enumerate(subs):
            j = np.array([0, s_.n_tile // 2, s_.n - 1])
            assert np.array_equal(merged.a["meta"][index[k][j].astype(np.int64)], parts[k].a["meta"][j])
        table_bytes = int(merged.n) * 68 + int(merged.ent_used) * 18
        gather = {"gather_ms": d2h_ms + merge_ms, "d2h_ms": d2h_ms, "merge_ms": merge_ms, "merge_kernel_ms": merge_kernel_ms,
                  "host_merge_ms": host_merge_ms, "rows": int(merged.n), "entries_before_compaction": int(hmerged.ent_used),
                  "entries": int(merged.ent_used), "table_bytes": table_bytes, "d2h_GBps": table_bytes / (d2h_ms * 1e-3) / 1e9,
                  "lines_per_s": n / ((d2h_ms + merge_ms) * 1e-3),
                  "what": "fg_merge_tables_device (rows back to their arrival positions, entries compacted into arrival order, in HBM) + D2H of the ONE merged table "
                          "into pinned memory (whole resident batch); host_merge_ms = fg_merge_tables on the host's cores, what rounds 2-3 "
                          "added to a D2H of the same size"}
        del hmerged
        if D.on and getattr(D, "threads", False):
            gather["ranks_what"] = "threads launcher: the ranks' tables are in one address space -- no rank gather"
        elif D.on:  # N ranks: the ranks' merged tables -> one table on rank 0, in rank (= shard) order
            D.barrier()
            t0 = time.perf_counter()
            full = shard.gather_distributed(merged, dst=0)
            D.barrier()
            gather["ranks_gather_ms"] = D.max((time.perf_counter() - t0) * 1e3)
            gather["ranks_what"] = "shard.gather_distributed: every rank's table to rank 0 (RCCL send / recv) + fg_gather_tables"
            if rank == 0:
                assert full.n == n * world
                gather["ranks_rows"] = int(full.n)
            gather["gather_ms"] += gather["ranks_gather_ms"]
        gather["all_ranks_gather_ms"] = [r[0] for r in D.all([gather["gather_ms"]])]
        del parts, merged

    # ---- PCIe-inclusive legs: every rank, at the same time -----------------------------------
    e2e = None
    if not args.no_e2e and wl in ("cfg2", "cfg1", "cfg3", "cfg4", "ltsv"):
        try:
            e2e = e2e_legs(D, dec, fmt, R.data, R.offsets, R.n_tile, R.tile_bytes, want_transcode=wl in ("cfg2", "cfg1"),
                           want_stream=wl in ("cfg2", "cfg1", "cfg3", "cfg4", "ltsv"))
        except AssertionError:
            raise
        except Exception as e:  # the PCIe legs never take the bench line down
            e2e = {"error": repr(e)}

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
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
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
                "kernel": KERNELS[fmt] if wl != "cfg5mix" else "fg::k_rfc5424 + fg::k_ltsv", "kernel_ms": kernel_ms,
                "algorithmic_bytes_per_launch": alg_read + alg_written,
                "moved_bytes_per_launch": moved, "moved_frac": moved / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                "read_only_frac": alg_read / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
            },
            "ranks": {"n": len(rank_ms), "kernel_ms": rank_ms, "kernel_ms_min": min(rank_ms), "kernel_ms_max": max(rank_ms),
                      "numa": numa,
                      "launcher": "threads (one process, one thread and ctx per GPU)" if thread_rank is not None else
                                  "self (torch.distributed.run)" if os.environ.get("FG_BENCH_SPAWNED") else
                                  "external (WORLD_SIZE set)" if "WORLD_SIZE" in os.environ else "single process"},
        }
        if opts:
            out["config"]["launch_opts"] = opts
        out["config"]["parity_checked_by"] = (
            "in this run: every replica's fixed columns == replica 0's, and the Ok count == the oracle's on the tile (cpu_baseline leg); the "
            "byte-exact Record comparison with the oracle is the -m gpu test suite's (tiles <= 250 K lines, tests/test_gpu_parity.py), "
            "not this script's")
        if calib is not None and "copy_GBps" in calib:
            rf = out["roofline"]
            rf["copy_GBps"], rf["read_GBps"] = calib["copy_GBps"], calib["read_GBps"]
            rf["copy_variants_GBps"] = {k: calib[k] for k in ("copy_gridstride_GBps", "copy_nt_GBps", "copy_flat_GBps")}
            rf["frac_of_copy"] = achieved / calib["copy_GBps"]
            rf["read_only_frac_of_read"] = alg_read / (kernel_ms * 1e-3) / 1e9 / calib["read_GBps"]
            rf["calibration"] = ("fg_calibrate_device in this process over the decoder's own resident buffer: the best of three float4 copies "
                                 "(grid-stride, non-temporal, one element per thread; 2 x bytes of traffic) and a read-only sweep, best of 3 "
                                 "launches each; frac_of_copy = achieved / copy_GBps -- the box-independent figure")
            if calib_all and len(calib_all) > 1:
                rf["per_rank_copy_GBps"] = [r[0] for r in calib_all]
        elif calib is not None:
            out["roofline"]["calibration_error"] = calib.get("error")
        if wl == "cfg5":
            out["roofline"]["frac_kind"] = ("EFFECTIVE, not HBM utilisation: heads only are staged (the kernel fetches ~0.58x the algorithmic "
                                            "bytes, profiles/traffic.json), so algorithmic bytes / time overstates what crosses HBM")
        if wl == "cfg5mix":
            a_ms = float(np.mean([ev[i][0].elapsed_time(sub_ev[i]) for i in range(args.steps)]))
            out["sub_batches"] = [
                {"format": "rfc5424", "lines": subs[0].n, "bytes": subs[0].tile_bytes * reps, "kernel_ms": a_ms,
                 "lines_per_s": subs[0].n / (a_ms * 1e-3)},
                {"format": "ltsv", "lines": subs[1].n, "bytes": subs[1].tile_bytes * reps, "kernel_ms": kernel_ms - a_ms,
                 "lines_per_s": subs[1].n / ((kernel_ms - a_ms) * 1e-3)}]
            out["gather_ms"] = gather["gather_ms"]
            out["gather"] = gather
        if wl == "cfg1":
            ems = float(np.mean([a.elapsed_time(b) for a, b in encode_ms[-args.steps:]]))
            out["encode"] = {"ms": ems, "out_bytes": enc_bytes, "lines_per_s": n / (ems * 1e-3),
                             "GBps_in_plus_out": (tile_bytes * reps + 68 * n + enc_bytes) / (ems * 1e-3) / 1e9,
                             "note": "fg_encode_device (count + scan + write kernels incl. the host sync for the total); "
                                     "value / ms_per_step cover decode + encode; roofline.* is the decode kernel alone"}
            dms = kernel_ms - ems  # roofline.* = the decode kernel alone, every figure of it (two were decode + encode)
            rf = out["roofline"]
            rf["kernel_ms"] = dms
            rf["achieved"] = (alg_read + alg_written) / (dms * 1e-3) / 1e9
            rf["frac"] = rf["achieved"] / HBM_PEAK_GBPS
            rf["moved_frac"] = moved / (dms * 1e-3) / 1e9 / HBM_PEAK_GBPS
            rf["read_only_frac"] = alg_read / (dms * 1e-3) / 1e9 / HBM_PEAK_GBPS
            if "copy_GBps" in rf:
                rf["frac_of_copy"] = rf["achieved"] / rf["copy_GBps"]
                rf["read_only_frac_of_read"] = alg_read / (dms * 1e-3) / 1e9 / rf["read_GBps"]
        if frame_ms is not None:
            out["framing"] = {"ms": frame_ms, "GBps": tile_bytes * reps / (frame_ms * 1e-3) / 1e9,
                              "note": "fg_frame_device: the one-pass framing scan (chained look-back over 128 KiB tiles; the stream is read once) incl. clearing the verdicts and the host sync that returns the frame count",
                              "fused": fused}
        # HBM traffic from the PMC passes (tools/prof.sh -> profiles/traffic.json): only when it was measured on THESE
        # kernel sources -- a figure from older code is not reported
        t = stamped(args.workload)
        if t:
            out["roofline"]["traffic"] = t["hbm_bytes_per_line"] * n
            out["roofline"]["traffic_profile"] = t.get("profile")
            cr = compute_roof(t)
            if cr:
                out["roofline"].update({k: cr[k] for k in ("valu_per_line", "salu_per_line", "valu_busy")})
                out["roofline"]["compute_what"] = cr["what"]
        elif (ROOT / "profiles" / "traffic.json").exists():
            out["roofline"]["traffic_note"] = "profiles/traffic.json holds no figure for these kernel sources: not reported"
        if e2e is not None:
            out["e2e"] = e2e
        extra_legs = []
        if wl == "cfg2" and world == 1 and not args.no_legs:
            # BASELINE configs[2] (GELF) and configs[3] (RFC5424 + structured data) on bounded samples, in this process
            for key, lfmt, desc, gen, skey in (
                    ("configs2", 2, WORKLOADS["cfg3"][1], lambda: synth.gelf_lines(250_000, invalid_frac=args.invalid_frac), "cfg3"),
                    ("configs3", 0, WORKLOADS["cfg4"][1], lambda: synth.rfc5424_lines(250_000, cfg=4, sd=True, invalid_frac=args.invalid_frac), "cfg4")):
                try:
                    # (16 M lines, 10 steps: at 4 M lines and 5 steps a 2 % kernel change drowned in the box's noise -- VERDICT r5 item 5)
                    out[key], leg = bounded_leg(desc, lfmt, gen(), 64, dev, local, D=D if not args.no_e2e else None, stamp_key=skey)
                    extra_legs.append((key, leg))
                except Exception as e:  # noqa: BLE001 -- an extra leg never takes the bench line down (a parity failure does: below)
                    if isinstance(e, AssertionError):
                        raise
                    out[key] = {"error": repr(e)[:200]}
        if wl == "cfg2" and world == 1 and not args.no_legs:
            try:
                out["small_batch"] = small_batch_leg(dev, local, args.invalid_frac)
            except Exception as e:  # noqa: BLE001 -- an extra leg never takes the bench line down
                if isinstance(e, AssertionError):
                    raise
                out["small_batch"] = {"error": repr(e)[:200]}
        if wl == "cfg2" and world == 1 and "WORLD_SIZE" not in os.environ and not args.no_mix:
            # BASELINE configs[4] (mixed RFC5424 + LTSV long-tail stream, host-side ordered gather) on a bounded sample, so that the
            # driver's default run carries its rate and gather time too: this very script, --workload cfg5mix, as a child process
            try:
                r = subprocess.run([sys.executable, str(Path(__file__).resolve()), "--workload", "cfg5mix", "--tile-lines", "200000", "--reps", "5",
                                    "--steps", "5", "--warmup", "2", "--no-cpu-baseline"], capture_output=True, text=True, timeout=300)
                m = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
                out["configs4"] = {"workload": m["config"]["workload"], "value": m["value"], "unit": m["unit"], "ms_per_step": m["ms_per_step"],
                                   "gather_ms": m["gather_ms"], "gather": m["gather"], "sub_batches": m["sub_batches"],
                                   "roofline_frac_effective": m["roofline"]["frac"],
                                   "roofline_frac_note": "EFFECTIVE: algorithmic bytes / time / 8 TB/s -- the long-tail kernels fetch the HEADS of the lines only "
                                                         "(0.61x the algorithmic bytes, profiles/traffic.json cfg5), so this is not an HBM utilisation",
                                   "what": "python bench.py --workload cfg5mix on a bounded sample (1 M lines resident); full size: that command alone"}
            except Exception as e:  # noqa: BLE001 -- the extra leg never takes the bench line down
                out["configs4"] = {"error": repr(e)[:200]}
        if world == 1 and not args.no_cpu_baseline:  # (the contract: the CPU leg runs on rank 0 at N = 1 only)
            os.sched_setaffinity(0, all_cpus)  # (the CPU leg uses every host core, not only the GPU's NUMA node)
            legs = [(s.fmt, s.data, s.offsets, s.n_tile, synth.LTSV_CONFIG if s.fmt == 1 else None) for s in subs]
            cb, n_ok_cpu = cpu_baseline(legs, pipeline=wl == "cfg1")
            assert n_ok_cpu == n_ok_tile, f"GPU Ok count {n_ok_tile} != oracle Ok count {n_ok_cpu}"
            out["cpu_baseline"] = cb
            for key, (lfmt, ldata, loffs, _ln, lcfg) in extra_legs:  # the bounded legs' Ok counts against the oracle (one plain pass)
                ok_cpu = oracle_ok_count(lfmt, ldata, loffs, lcfg)
                assert ok_cpu == out[key]["ok_lines_per_tile"], f"{key}: GPU Ok count {out[key]['ok_lines_per_tile']} != oracle Ok count {ok_cpu}"
                out[key]["ok_count_checked_against"] = "oracle"
        print(json.dumps(out), flush=True)
    D.close()


if __name__ == "__main__":
    main()
