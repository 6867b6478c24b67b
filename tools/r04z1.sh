#!/bin/bash
T=${1:-r04z1}
mkdir -p gpurun_out
export FG_BENCH_CACHE=/tmp/fg_bench_cache
timeout 600 python -m pytest tests -m gpu -x -q -k "ltsv or LTSV" > gpurun_out/${T}_pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/${T}_pytest.log
timeout 300 python tools/sweep.py ltsv --lines 250000 --reps 16 ";tile_cap=15360;tile_cap=16384;tile_cap=17408;chunk_lines=512;chunk_lines=1024" 2>/dev/null | grep "M lines/s" | tee gpurun_out/${T}_sweep_ltsv.log
python bench.py --workload ltsv --tile-lines 250000 --reps 400 --steps 10 --warmup 2 --no-e2e --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${T}_bench_ltsv_100M.json
python bench.py --workload ltsv5 --tile-lines 250000 --reps 80 --steps 5 --warmup 1 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 > gpurun_out/${T}_bench_ltsv5_20M.json
for f in bench_ltsv_100M bench_ltsv5_20M; do python -c "
import json; d=json.loads(open('gpurun_out/${T}_$f.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$f', round(d['value']/1e6,1), 'M lines/s', round(r['kernel_ms'],3), 'ms frac', round(r['frac'],4))" 2>&1 | tail -1; done
