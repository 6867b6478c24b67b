#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r02u.log; : > $O
run() { echo "== $*" >> $O; env "$@" FG_PLAN=1 python bench.py --workload cfg3 --steps 10 --warmup 2 --no-cpu-baseline --no-e2e 2> gpurun_out/err.txt | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value']/1e6, d['roofline'].get('kernel_ms'), d['roofline'].get('frac'))" >> $O; grep -m1 "gelf plan" gpurun_out/err.txt >> $O; }
run FG_LINES_PER_GROUP=16 FG_GELF_WINDOW=5
run FG_LINES_PER_GROUP=16 FG_GELF_WINDOW=6
run FG_LINES_PER_GROUP=12 FG_GELF_WINDOW=4
run FG_LINES_PER_GROUP=32 FG_GELF_WINDOW=6
FG_LINES_PER_GROUP=16 FG_GELF_WINDOW=5 FG_PROF=1 python bench.py --workload cfg3 --steps 1 --warmup 1 --no-cpu-baseline --no-e2e 2>&1 | grep -A1 -m1 "prof" >> $O
cat $O
