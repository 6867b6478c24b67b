"""GPU: the boundary's concurrency contract and the bench launcher (round 2).

  * one decoder clone per connection thread (`Send`, not `Sync`): src/flowgger/input/tcp/tcp_input.rs:39-47,
    decoder/mod.rs:29-36 -- N host threads, each with its own fg_clone, decode different batches concurrently on one
    device; every result must equal the oracle's
  * `python bench.py --gpus N` launches its own ranks: the self-spawn path is exercised with one rank"""
import json
import subprocess
import sys
import threading
from pathlib import Path

import numpy as np
import pytest

from flowgger_amd import GelfDecoder, LTSVDecoder, RFC3164Decoder, RFC5424Decoder, synth, tzdb
from golden.reference_vectors import GELF, LTSV, RFC3164, RFC5424

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_eight_threads_each_with_its_own_clone(oracle):
    oracle.set_rfc3164(2026, tzdb.default_table())
    protos = {
        RFC5424: (RFC5424Decoder(), None),
        LTSV: (LTSVDecoder(synth.LTSV_CONFIG), synth.LTSV_CONFIG),
        GELF: (GelfDecoder(), None),
        RFC3164: (RFC3164Decoder({"rfc3164": {"current_year": 2026}}), None),
    }

    def corpus(fmt, seed):
        n = 20_000 + 1000 * seed
        if fmt == RFC5424:
            return synth.rfc5424_lines(n, cfg=40 + seed, sd=bool(seed % 8 >= 4))  # (cfg only seeds the generator)
        if fmt == LTSV:
            return synth.ltsv_lines(n, cfg=50 + seed)
        if fmt == GELF:
            return synth.gelf_lines(n, cfg=60 + seed)
        return synth.rfc3164_lines(n, cfg=70 + seed)

    jobs = []
    for k in range(8):
        fmt = (RFC5424, LTSV, GELF, RFC3164)[k % 4]
        proto, cfg = protos[fmt]
        data, offsets = synth.pack(corpus(fmt, k))
        want = oracle.decode_batch(fmt, data, offsets, cfg)
        jobs.append((proto.clone_boxed(), fmt, data, offsets, want))  # the clone is made on the spawning thread, as tcp_input does
    errors = []
    start = threading.Barrier(len(jobs))

    def worker(k):
        dec, fmt, data, offsets, (oblob, ooffs) = jobs[k]
        try:
            start.wait()
            for _ in range(6):  # several batches per "connection": buffers of a ctx are reused across calls
                tab = dec.decode_packed(data, offsets)
                blob, offs = tab.serialize(fmt, data, offsets, cfg=dec._cfg)
                if not (np.array_equal(offs, ooffs) and np.array_equal(blob, oblob)):
                    errors.append(f"thread {k} (format {fmt}): result differs from the oracle")
                    return
        except Exception as e:  # noqa: BLE001
            errors.append(f"thread {k}: {e!r}")

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(len(jobs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    # the prototypes are still usable after their clones are gone
    for dec, *_ in jobs:
        dec.close()
    proto, _ = protos[RFC5424]
    assert proto.decode("<13>1 2015-08-05T15:53:45Z h a p m - x").msg == "x"


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 1 --spawn`: the torch.distributed.run self-launch that `--gpus N>1` takes when no launcher
    set WORLD_SIZE, with one rank (this box has one GPU)."""
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--spawn", "--steps", "2", "--warmup", "1", "--tile-lines", "50000",
           "--reps", "4", "--no-cpu-baseline", "--no-e2e"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["value"] > 0
    assert out["ranks"]["launcher"].startswith("self") and out["ranks"]["n"] == 1
    assert out["config"]["lines_per_gpu"] == 200000
