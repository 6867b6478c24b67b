#!/bin/bash
mkdir -p gpurun_out
: > gpurun_out/r02d_cfg3.log
for L in ${LS:-8 16 32}; do
  echo "## L=$L" >> gpurun_out/r02d_cfg3.log
  FG_LINES_PER_GROUP=$L python bench.py --workload cfg3 --tile-lines 200000 --reps 20 --steps 5 --warmup 1 --no-cpu-baseline --no-e2e 2>> gpurun_out/r02d_cfg3.err | tail -1 | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); r=d['roofline']; print(json.dumps({'Mlines_s':round(d['value']/1e6,1),'kernel_ms':round(r['kernel_ms'],3),'frac':round(r['frac'],4)}))" >> gpurun_out/r02d_cfg3.log
  FG_PROF=1 FG_LINES_PER_GROUP=$L python bench.py --workload cfg3 --tile-lines 200000 --reps 20 --steps 2 --warmup 1 --no-cpu-baseline --no-e2e 2>&1 | grep "fg prof" | tail -2 >> gpurun_out/r02d_cfg3.log
done
cat gpurun_out/r02d_cfg3.log
