#!/bin/bash
# the PCIe-inclusive legs of tools/r05_final.sh again (its first run handed the harness PAGEABLE memory beyond the pinned allocator's
# 1 GiB cap of that commit: the raw-stream legs measured staged copies): the default bench line + the entry corpora's legs + cfg1
T=${1:-r05fin2}
mkdir -p gpurun_out
export FG_BENCH_CACHE=/tmp/fg_bench_cache
env -u FG_BENCH_CACHE python bench.py 2> gpurun_out/${T}_bench_default.err | tail -1 > gpurun_out/${T}_bench_default_100M.json
for w in cfg3 ltsv cfg4; do python bench.py --workload $w --tile-lines 250000 --reps 16 --steps 5 --warmup 2 --no-cpu-baseline --no-calib 2>/dev/null | tail -1 > gpurun_out/${T}_e2e_$w.json; done
python bench.py --workload cfg1 --reps 4 --steps 5 --warmup 2 2>/dev/null | tail -1 > gpurun_out/${T}_bench_cfg1_pipeline.json
for f in bench_default_100M e2e_cfg3 e2e_ltsv e2e_cfg4 bench_cfg1_pipeline; do python -c "
import json; d=json.loads(open('gpurun_out/${T}_$f.json').read().strip().splitlines()[-1]); r=d['roofline']; e=d.get('e2e') or {}
print('$f', round(d['value']/1e6,1), 'M lines/s', round(r['kernel_ms'],3), 'ms frac', round(r['frac'],4), 'of copy', r.get('frac_of_copy'), d.get('encode',{}).get('ms'), {k: round(v/1e6,1) for k,v in (e.get('aggregate') or {}).items()}, {k: round(v['lines_per_s']/1e6,1) for k,v in e.items() if isinstance(v,dict) and 'lines_per_s' in v})" 2>&1 | tail -1; done
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r05fin2_bench_default_100M.json').read())
for k in ('configs2', 'configs3', 'configs4'):
    c = d.get(k, {})
    print(k, round(c.get('value', 0) / 1e9, 3), 'G', c.get('roofline_frac'), c.get('read_only_frac'), c.get('gather_ms'), {kk: round(v['lines_per_s'] / 1e6, 1) for kk, v in (c.get('e2e') or {}).items() if isinstance(v, dict) and 'lines_per_s' in v})
print('small_batch', {k: {n: round(v['lines_per_s'] / 1e6, 1) for n, v in vv.items()} for k, vv in d.get('small_batch', {}).items() if isinstance(vv, dict)})
PY
