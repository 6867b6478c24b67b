#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r02z.log; : > $O
for ab in 2 1 8; do
  FG_PROF=1 FG_ABLATE=$ab bash tools/prof_traffic.sh ltsv_ab$ab k_ltsv --workload ltsv --tile-lines 200000 --reps 20 > /dev/null 2>&1
  python - >> $O <<PY
import json
w='ltsv_ab$ab'
s=json.load(open(f'gpurun_out/traffic_{w}.json')); n=4000000; h=s['hbm_bytes_per_dispatch']
print(w, 'per line: total', h['total']/n, 'read', h['read']/n, 'written', h['written']/n)
PY
done
cat $O
