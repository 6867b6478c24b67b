#!/bin/bash
T=${1:-r04n}
mkdir -p gpurun_out
export FG_BENCH_CACHE=/tmp/fg_bench_cache
timeout 900 python -m pytest tests/test_gpu_round4.py -m gpu -x -q -k "zero_copy" > gpurun_out/${T}_pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/${T}_pytest.log
env -u FG_BENCH_CACHE timeout 400 python bench.py 2> gpurun_out/${T}_bench_default.err | tail -1 > gpurun_out/${T}_bench_default_100M.json
python - <<PY
import json
d=json.loads(open("gpurun_out/${T}_bench_default_100M.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("cfg2", round(d["value"]/1e9,2), "G", round(r["frac"],4), r.get("copy_variants_GBps"), round(r.get("read_GBps",0)), round(r.get("frac_of_copy",0),3), "traffic", r.get("traffic"))
for k in ("configs2","configs3","configs4"):
    c=d[k]; print(k, round(c["value"]/1e6,1), round(c.get("roofline_frac",0),4), {kk:(round(vv["lines_per_s"]/1e6,1)) for kk,vv in (c.get("e2e") or {}).items() if isinstance(vv,dict) and "lines_per_s" in vv})
print({k: round(v/1e6,1) for k,v in d["e2e"]["aggregate"].items()})
PY
for w in cfg3 ltsv cfg4; do python bench.py --workload $w --tile-lines 250000 --reps 16 --steps 5 --warmup 2 --no-cpu-baseline --no-calib 2>/dev/null | tail -1 > gpurun_out/${T}_e2e_$w.json; python -c "
import json; d=json.loads(open('gpurun_out/${T}_e2e_$w.json').read().strip().splitlines()[-1]); e=d['e2e']; print('$w', {k: round(v/1e6,1) for k,v in e['aggregate'].items()}, 'traffic', d['roofline'].get('traffic'))"; done
