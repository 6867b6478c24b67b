#!/bin/bash
T=${1:-r04z4}
mkdir -p gpurun_out
export FG_BENCH_CACHE=/tmp/fg_bench_cache
timeout 600 python -m pytest tests -m gpu -x -q -k "ltsv or LTSV" > gpurun_out/${T}_pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/${T}_pytest.log
for i in 1 2; do
python bench.py --workload ltsv --tile-lines 250000 --reps 400 --steps 10 --warmup 2 --no-e2e --no-cpu-baseline --no-calib 2>/dev/null | tail -1 > gpurun_out/${T}_bench_ltsv_100M.json
python -c "
import json; d=json.loads(open('gpurun_out/${T}_bench_ltsv_100M.json').read().strip().splitlines()[-1]); r=d['roofline']
print('ltsv 100M', round(d['value']/1e6,1), 'M lines/s', round(r['kernel_ms'],3), 'ms')"
done
timeout 300 python tools/sweep.py ltsv --lines 250000 --reps 400 ";chunk_lines=256;;chunk_lines=256" 2>/dev/null | grep "M lines/s" | tee gpurun_out/${T}_sweep_ltsv_100M.log
