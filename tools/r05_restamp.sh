#!/bin/bash
# after the last edit of a kernel source (the launch policies moved into fg_plan_policy.hpp: same arithmetic, new source hash): the GPU
# suite on the final tree, the HBM-traffic passes of every workload again (profiles/traffic.json is keyed on the kernels' sources), the
# default bench line
T=${1:-r05z}
mkdir -p gpurun_out
export FG_BENCH_CACHE=/tmp/fg_bench_cache
python -m pytest tests -m gpu -x -q > gpurun_out/${T}_gpu_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${T}_gpu_pytest.log
tail -3 gpurun_out/${T}_gpu_pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
PAT=k_rfc5424 bash tools/prof.sh ${T}_cfg2 --reps 100 --no-mix --no-legs --no-calib > gpurun_out/${T}_prof_cfg2.log 2>&1
bash tools/prof_traffic.sh ${T}_cfg3 'k_gelf<' --workload cfg3 --tile-lines 250000 --reps 16 --no-calib > /dev/null 2>&1
bash tools/prof_traffic.sh ${T}_cfg4 k_rfc5424 --workload cfg4 --tile-lines 250000 --reps 16 --no-calib > /dev/null 2>&1
bash tools/prof_traffic.sh ${T}_cfg5 k_rfc5424 --workload cfg5 --tile-lines 250000 --reps 16 --no-calib > /dev/null 2>&1
bash tools/prof_traffic.sh ${T}_ltsv k_ltsv --workload ltsv --tile-lines 250000 --reps 16 --no-calib > /dev/null 2>&1
bash tools/prof_traffic.sh ${T}_ltsv5 k_ltsv --workload ltsv5 --tile-lines 250000 --reps 16 --no-calib > /dev/null 2>&1
env -u FG_BENCH_CACHE python bench.py 2> gpurun_out/${T}_bench_default.err | tail -1 > gpurun_out/${T}_bench_default_100M.json
python - <<PY
import json
d = json.loads(open('gpurun_out/${T}_bench_default_100M.json').read())
r = d['roofline']
print('headline', round(d['value'] / 1e9, 2), 'G lines/s frac', round(r['frac'], 3), 'of copy', round(r.get('frac_of_copy', 0), 3), 'traffic', r.get('traffic'), r.get('traffic_note'))
for k in ('configs2', 'configs3', 'configs4'):
    c = d.get(k, {})
    print(k, round(c.get('value', 0) / 1e9, 3), 'G', c.get('roofline_frac'), c.get('gather_ms'), {kk: round(v['lines_per_s'] / 1e6, 1) for kk, v in (c.get('e2e') or {}).items() if isinstance(v, dict) and 'lines_per_s' in v})
print('small_batch', {k: {n: round(v['lines_per_s'] / 1e6, 1) for n, v in vv.items()} for k, vv in d.get('small_batch', {}).items() if isinstance(vv, dict)})
print('e2e', {k: round(v['lines_per_s'] / 1e6, 1) for k, v in (d.get('e2e') or {}).items() if isinstance(v, dict) and 'lines_per_s' in v})
PY
