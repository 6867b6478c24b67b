#!/bin/bash
# round 5, call 2: the new GPU tests first, then the whole GPU suite; then static round-robin vs ticket dispatch on ONE box --
# small batches (tools/probe/small_batch.py) and 4 M-line batches (tools/sweep.py) -- and a kernel trace of GELF at the sizes where
# its time is not monotonic in the batch size (profiles/r05a_small.log: 256 K lines 898 us, 512 K lines 461 us)
T=${1:-r05b}
mkdir -p gpurun_out
export FG_BENCH_CACHE=/tmp/fg_bench_cache
timeout 900 python -m pytest tests/test_gpu_round5.py -x -q > gpurun_out/${T}_gpu_pytest_round5.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${T}_gpu_pytest_round5.log
tail -5 gpurun_out/${T}_gpu_pytest_round5.log
timeout 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_round5.py > gpurun_out/${T}_gpu_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${T}_gpu_pytest.log
tail -5 gpurun_out/${T}_gpu_pytest.log
export FG_PROBE_SIZES=16384,65536,262144,524288,1048576
FG_PROBE_OPTS=';static_chunks=1' python tools/probe/small_batch.py cfg2 cfg5 ltsv5 cfg4 cfg3 ltsv > gpurun_out/${T}_small_ab.log 2>&1
grep -h "n=" gpurun_out/${T}_small_ab.log
for w in cfg5 cfg4 ltsv5 cfg3; do python tools/sweep.py $w --lines 250000 --reps 16 ';static_chunks=1;;static_chunks=1' 2>&1 | grep "M lines/s"; done | tee gpurun_out/${T}_sweep_4M.log
python tools/sweep.py cfg2 --lines 1000000 --reps 40 ';static_chunks=1;;static_chunks=1' 2>&1 | grep "M lines/s" | tee gpurun_out/${T}_sweep_cfg2_40M.log
cd /tmp && export TMPDIR=/tmp
FG_PROBE_SIZES=65536,262144,524288 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/${T}_gelf_trace -o gelf -- python /root/repo/tools/probe/small_batch.py cfg3 > /root/repo/gpurun_out/${T}_gelf_trace.log 2>&1
cd /root/repo
python - <<'PY'
import csv, glob, collections
for f in glob.glob("gpurun_out/r05b_gelf_trace/**/*kernel_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    by = collections.defaultdict(list)
    for r in rows:
        by[(r["Kernel_Name"][:60], r.get("Grid_Size_X") or r.get("Grid_Size"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in sorted(by.items()):
        print(k, len(v), "launches, median us", sorted(v)[len(v) // 2])
PY
