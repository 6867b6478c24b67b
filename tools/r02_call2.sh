#!/bin/bash
# round 2, GPU call 2: wave-cooperative GELF kernel -- parity suite, then cfg3 timings over lines-per-group
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -k "gelf or Gelf or entry_workloads or reference_vectors or transcode or encoder" > gpurun_out/r02c_pytest_gelf.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02c_pytest_gelf.log
tail -5 gpurun_out/r02c_pytest_gelf.log
: > gpurun_out/r02c_cfg3.log
for L in 16 32 64; do
  echo "## L=$L" >> gpurun_out/r02c_cfg3.log
  FG_LINES_PER_GROUP=$L python bench.py --workload cfg3 --tile-lines 200000 --reps 20 --steps 5 --warmup 1 --no-cpu-baseline --no-e2e 2>> gpurun_out/r02c_cfg3.err | tail -1 | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); r=d['roofline']; print(json.dumps({'Mlines_s':round(d['value']/1e6,1),'kernel_ms':round(r['kernel_ms'],3),'frac':round(r['frac'],4)}))" >> gpurun_out/r02c_cfg3.log
  FG_PROF=1 FG_LINES_PER_GROUP=$L python bench.py --workload cfg3 --tile-lines 200000 --reps 20 --steps 2 --warmup 1 --no-cpu-baseline --no-e2e 2>&1 | grep "fg prof" | tail -1 >> gpurun_out/r02c_cfg3.log
done
cat gpurun_out/r02c_cfg3.log
FG_LINES_PER_GROUP=16 bash tools/prof.sh r02c_cfg3 --workload cfg3 --tile-lines 200000 --reps 20 > gpurun_out/r02c_prof_cfg3.log 2>&1
tail -3 gpurun_out/r02c_prof_cfg3.log
