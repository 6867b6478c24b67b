#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r02ad.log; : > $O
run() { w=$1; shift; echo "== $w $*" >> $O; env "$@" python bench.py --workload $w --tile-lines 200000 --reps 20 --steps 5 --warmup 1 --no-cpu-baseline --no-e2e 2> gpurun_out/err.txt | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value']/1e6, d['roofline'].get('kernel_ms'), d['roofline'].get('frac'))" >> $O; grep "fg prof" gpurun_out/err.txt | head -1 >> $O; }
run cfg4 A=1
run cfg4 FG_LINES_PER_GROUP=64
run cfg4 FG_LINES_PER_GROUP=64 FG_TILE_CAP=40960
run cfg4 FG_LINES_PER_GROUP=16
run cfg5 A=1
run cfg2 A=1
cat $O
