#!/bin/bash
# round 3, second GPU call: the framer / side-effect tests, the latency sweep, the CPU baseline's thread scaling on this box
T=${1:-r03b}
mkdir -p gpurun_out
python -m pytest tests/test_gpu_round3.py tests/test_gpu_parity.py -m gpu -x -q -k "round3 or ltsv or cpp_host or stall or micro or stdout" > gpurun_out/${T}_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${T}_pytest.log
tail -15 gpurun_out/${T}_pytest.log
python tools/host_path_bench.py --workload latency > gpurun_out/${T}_latency.log 2>&1; tail -12 gpurun_out/${T}_latency.log | cut -c1-200
echo "--- cpu"; nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /proc/loadavg; lscpu | grep -E "Model name|Socket|Core|Thread|NUMA node\(s\)"
python - <<'PY'
import sys, os
sys.path.insert(0, 'tests')
import oracle_binding
from flowgger_amd import synth
o = oracle_binding.Oracle()
data, offs = synth.pack(synth.rfc5424_lines(1_000_000, cfg=2))
base = None
for th in (1, 2, 4, 8, 16, 32, 64, 128, 256):
    if th > (os.cpu_count() or 1): break
    s, n = o.bench_timed(0, data, offs, th, 1.5)
    r = n / s
    base = base or r
    print(f"threads {th:4d}: {r/1e6:9.2f} M lines/s  speed-up {r/base:7.2f}  efficiency {r/base/th:5.2f}", flush=True)
PY
