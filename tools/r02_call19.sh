#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r02p.log; : > $O
run() { echo "== $*" >> $O; env "$@" FG_PLAN=1 python bench.py --workload cfg3 --steps 10 --warmup 2 --no-cpu-baseline --no-e2e 2> gpurun_out/err.txt | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value']/1e6, d['roofline'].get('kernel_ms'), d['roofline'].get('frac'))" >> $O; grep -m1 "gelf plan" gpurun_out/err.txt >> $O; }
run A=1
run FG_LINES_PER_GROUP=4
run FG_LINES_PER_GROUP=8 FG_GELF_W5=1
run FG_LINES_PER_GROUP=4 FG_GELF_W5=1
run FG_LINES_PER_GROUP=8 FG_TILE_CAP=3072
run FG_LINES_PER_GROUP=8 FG_WAVES_PER_CU=12
FG_PROF=1 python bench.py --workload cfg3 --steps 1 --warmup 1 --no-cpu-baseline --no-e2e 2>&1 | grep -A1 -m1 "prof" >> $O
for w in cfg4 cfg5; do python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$w', d['value']/1e6, d['roofline'].get('kernel_ms'), d['roofline'].get('frac'))" >> $O; done
cat $O
