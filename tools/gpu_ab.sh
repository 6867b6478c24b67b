#!/bin/bash
# usage: tools/gpu_ab.sh <outfile> -- runs bench.py under a list of env-knob settings (A/B on one box)
OUT=${1:-gpurun_out/ab.log}; shift
: > $OUT
while read -r envs; do
  [ -z "$envs" ] && continue
  echo "## $envs" >> $OUT
  env $envs python bench.py --steps 10 --warmup 2 --reps 40 --no-cpu-baseline --no-e2e $BENCH_ARGS 2> gpurun_out/ab_stderr.tmp | tail -1 | python -c "
import sys,json
for l in sys.stdin:
    try:
        d=json.loads(l); r=d['roofline']; print(json.dumps({'glines_s':round(d['value']/1e9,3),'kernel_ms':round(r['kernel_ms'],4),'frac':round(r['frac'],4)}))
    except Exception as e: print('ERR',l[:300])
" >> $OUT
  grep -h "fg prof" gpurun_out/ab_stderr.tmp | tail -1 >> $OUT
done
cat $OUT
