#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r02k_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02k_pytest.log
tail -4 gpurun_out/r02k_pytest.log
BENCH_ARGS="--no-cpu-baseline --no-e2e" bash tools/gpu_workloads.sh gpurun_out/r02k_workloads.log cfg4 cfg5 > /dev/null 2>&1
cat gpurun_out/r02k_workloads.log
for L in 8 16 32; do
echo "## SD L=$L" >> gpurun_out/r02k_workloads.log
FG_SD_LINES_PER_GROUP=$L python bench.py --workload cfg4 --tile-lines 200000 --reps 20 --steps 5 --warmup 1 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); r=d['roofline']; print(json.dumps({'Mlines_s':round(d['value']/1e6,1),'kernel_ms':round(r['kernel_ms'],3),'frac':round(r['frac'],4)}))" >> gpurun_out/r02k_workloads.log
done
tail -6 gpurun_out/r02k_workloads.log
python bench.py --steps 10 --warmup 2 --reps 40 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 | cut -c1-400
