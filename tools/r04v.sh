#!/bin/bash
# tables in LDS for the pair-parallel SD kernel: parity + cfg4 / cfg5 / cfg2 same-box numbers
T=${1:-r04v}
mkdir -p gpurun_out
export FG_BENCH_CACHE=/tmp/fg_bench_cache
timeout 600 python -m pytest tests -m gpu -x -q -k "sd or SD or cfg4 or cfg5 or whole_line" > gpurun_out/${T}_pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/${T}_pytest.log
timeout 300 python tools/sweep.py cfg4 --lines 250000 --reps 16 ";sd_walk=1" 2>/dev/null | grep "M lines/s" | tee gpurun_out/${T}_sweep_cfg4.log
python bench.py --workload cfg4 --tile-lines 250000 --reps 500 --steps 5 --warmup 1 --no-e2e --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${T}_bench_cfg4_125M.json
python bench.py --workload cfg5 --tile-lines 250000 --reps 160 --steps 5 --warmup 1 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 > gpurun_out/${T}_bench_cfg5_40M.json
python bench.py --no-legs --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 > gpurun_out/${T}_bench_cfg2.json
for f in bench_cfg4_125M bench_cfg5_40M bench_cfg2; do python -c "
import json; d=json.loads(open('gpurun_out/${T}_$f.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$f', round(d['value']/1e6,1), 'M lines/s', round(r['kernel_ms'],3), 'ms frac', round(r['frac'],4), 'of copy', r.get('frac_of_copy'))" 2>&1 | tail -1; done
