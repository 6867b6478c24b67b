#!/bin/bash
# usage: tools/gpu_workloads.sh <outfile> [workloads...] -- kernel timings of the non-headline corpora on one box
OUT=${1:-gpurun_out/workloads.log}; shift
WLS=${@:-cfg4 cfg5 cfg3 ltsv}
: > $OUT
for w in $WLS; do
  echo "## $w" >> $OUT
  python bench.py --workload $w --tile-lines 200000 --reps 20 --steps 5 --warmup 1 $BENCH_ARGS 2> gpurun_out/wl_stderr.tmp | tail -1 | python -c "
import sys,json
for l in sys.stdin:
    try:
        d=json.loads(l); r=d['roofline']; c=d.get('cpu_baseline',{})
        print(json.dumps({'Mlines_s':round(d['value']/1e6,1),'kernel_ms':round(r['kernel_ms'],3),'frac':round(r['frac'],4),'GBps':round(r['achieved'],1),'avgB':round(d['config']['bytes_per_gpu']/d['config']['lines_per_gpu'],1),'cpu_Mlines_s':round(c.get('value',0)/1e6,2),'cores':c.get('cores')}))
    except Exception as e: print('ERR',l[:300])
" >> $OUT
  tail -3 gpurun_out/wl_stderr.tmp | grep -v "^$" | cut -c1-300 >> $OUT
done
cat $OUT
