#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r02y.log; : > $O
run() { w=$1; shift; echo "== $w $*" >> $O; env "$@" python bench.py --workload $w --tile-lines 200000 --reps 20 --steps 5 --warmup 1 --no-cpu-baseline --no-e2e 2> gpurun_out/err.txt | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value']/1e6, d['roofline'].get('kernel_ms'), d['roofline'].get('frac'))" >> $O; }
run ltsv A=1
run cfg4 A=1
run cfg3 A=1
run cfg2 A=1
run rfc3164 A=1
python bench.py --workload cfg2 --steps 10 --warmup 2 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cfg2 100M', d['value']/1e6, d['roofline'].get('kernel_ms'), d['roofline'].get('frac'))" >> $O
bash tools/prof_traffic.sh ltsv k_ltsv --workload ltsv --tile-lines 200000 --reps 20 > /dev/null 2>&1
bash tools/prof_traffic.sh cfg4 k_rfc5424 --workload cfg4 --tile-lines 200000 --reps 20 > /dev/null 2>&1
python - >> $O <<'PY'
import json
for w in ('ltsv','cfg4'):
    s=json.load(open(f'gpurun_out/traffic_{w}.json')); b=json.loads(open(f'gpurun_out/traffic_{w}_bench.json').read().strip().splitlines()[-1])
    n=b['config']['lines_per_gpu']; h=s['hbm_bytes_per_dispatch']
    print(w, 'per line: total', h['total']/n, 'read', h['read']/n, 'written', h['written']/n, 'algorithmic', b['roofline']['algorithmic_bytes_per_launch']/n)
PY
cat $O
