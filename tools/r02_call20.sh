#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r02q.log; : > $O
python bench.py --workload cfg3 --steps 10 --warmup 2 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 | cut -c1-200 >> $O
FG_PROF=1 python bench.py --workload cfg3 --steps 1 --warmup 1 --no-cpu-baseline --no-e2e 2>&1 | grep -A1 -m1 "prof" >> $O
FG_PROF=1 FG_LINES_PER_GROUP=16 python bench.py --workload cfg3 --steps 1 --warmup 1 --no-cpu-baseline --no-e2e 2>&1 | grep -A1 -m1 "prof" >> $O
cat $O
