#!/bin/bash
# after the last change of fg_rfc5424.hip: what depends on it, measured again on one box
T=${1:-r03z}
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/${T}_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${T}_pytest.log
tail -3 gpurun_out/${T}_pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py 2> gpurun_out/${T}_bench_default.err | tail -1 > gpurun_out/${T}_bench_default_100M.json
cut -c1-200 gpurun_out/${T}_bench_default_100M.json
bash tools/prof.sh ${T}_cfg2 --reps 40 --no-mix > gpurun_out/${T}_prof_cfg2.log 2>&1
bash tools/prof_traffic.sh ${T}_cfg4 k_rfc5424 --workload cfg4 --tile-lines 1000000 --reps 4 > /dev/null 2>&1
bash tools/prof_traffic.sh ${T}_cfg5 k_rfc5424 --workload cfg5 --tile-lines 1000000 --reps 4 > /dev/null 2>&1
python bench.py --workload cfg4 --reps 125 --steps 5 --warmup 1 --no-e2e 2>/dev/null | tail -1 > gpurun_out/${T}_bench_cfg4_125M.json
python bench.py --workload cfg5 --reps 40 --steps 5 --warmup 1 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 > gpurun_out/${T}_bench_cfg5_40M.json
python bench.py --workload cfg5mix 2>/dev/null | tail -1 > gpurun_out/${T}_bench_cfg5mix.json
python bench.py --workload frame --steps 5 --warmup 1 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 > gpurun_out/${T}_bench_frame.json
python bench.py --workload cfg1 --reps 4 --steps 5 --warmup 2 2>/dev/null | tail -1 > gpurun_out/${T}_bench_cfg1_pipeline.json
for f in default_100M cfg4_125M cfg5_40M cfg5mix frame cfg1_pipeline; do python -c "
import json; d=json.loads(open('gpurun_out/${T}_bench_$f.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$f', round(d['value']/1e6,1), 'M lines/s', round(r['kernel_ms'],3), 'ms frac', round(r['frac'],4), 'cpu', d.get('cpu_baseline',{}).get('value'), d.get('gather_ms'), d.get('framing',{}).get('GBps'), d.get('encode',{}).get('ms'))" 2>&1 | tail -1; done
