#!/bin/bash
# GELF rows (one line per row of 16 lanes, verdicts by row reductions) + LTSV scratch: GPU suite, cfg3 / ltsv at full size, counters
T=${1:-r04q}
mkdir -p gpurun_out
export FG_BENCH_CACHE=/tmp/fg_bench_cache
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_gpu_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/${T}_gpu_pytest.log
python bench.py --workload cfg3 --tile-lines 250000 --reps 400 --steps 10 --warmup 2 --no-e2e --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${T}_bench_cfg3_100M.json
python bench.py --workload ltsv --tile-lines 250000 --reps 400 --steps 10 --warmup 2 --no-e2e --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${T}_bench_ltsv_100M.json
PAT='k_gelf<' bash tools/prof_quick.sh ${T}_cfg3 'k_gelf<' --workload cfg3 --tile-lines 250000 --reps 16 --no-calib > /dev/null 2>&1
for f in bench_cfg3_100M bench_ltsv_100M; do python -c "
import json; d=json.loads(open('gpurun_out/${T}_$f.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$f', round(d['value']/1e6,1), 'M lines/s', round(r['kernel_ms'],3), 'ms frac', round(r['frac'],4), 'of copy', r.get('frac_of_copy'))" 2>&1 | tail -1; done
python - <<PY
import json
d=json.load(open("gpurun_out/prof_${T}_cfg3.json"))
print(json.dumps(d, indent=0)[:1500])
PY
