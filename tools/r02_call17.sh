#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r02n_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02n_pytest.log
tail -4 gpurun_out/r02n_pytest.log
BENCH_ARGS="--no-cpu-baseline --no-e2e" bash tools/gpu_workloads.sh gpurun_out/r02n_workloads.log rfc3164 > /dev/null 2>&1
cat gpurun_out/r02n_workloads.log
python bench.py --workload rfc3164 --steps 10 --warmup 2 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 | cut -c1-330
python bench.py --workload cfg1 --reps 4 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r02n_cfg1.json
python -c "
import json; d=json.loads(open('gpurun_out/r02n_cfg1.json').read().strip().splitlines()[-1]); print('cfg1', d['value']/1e6, d.get('encode',{}).get('ms'), json.dumps(d.get('e2e'))[:600])"
