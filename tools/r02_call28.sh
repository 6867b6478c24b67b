#!/bin/bash
mkdir -p gpurun_out
bash tools/prof_traffic.sh ltsv k_ltsv --workload ltsv --tile-lines 200000 --reps 20 2>&1 | tail -1 | cut -c1-600
bash tools/prof_traffic.sh cfg4 k_rfc5424 --workload cfg4 --tile-lines 200000 --reps 20 2>&1 | tail -1 | cut -c1-600
python - <<'PY'
import json
for w in ('ltsv','cfg4'):
    s=json.load(open(f'gpurun_out/traffic_{w}.json')); b=json.loads(open(f'gpurun_out/traffic_{w}_bench.json').read().strip().splitlines()[-1])
    n=b['config']['lines_per_gpu']; h=s['hbm_bytes_per_dispatch']
    print(w, 'per line: total', h['total']/n, 'read', h['read']/n, 'written', h['written']/n, 'algorithmic', b['roofline']['algorithmic_bytes_per_launch']/n)
PY
