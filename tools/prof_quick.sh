#!/bin/bash
# usage: tools/prof_quick.sh <tag> <kernel-substring> [bench args...] -- kernel trace + ONE PMC pass (SQ counters) of bench.py;
# condensed summary in gpurun_out/prof_<tag>.json (tools/prof.sh does the full set incl. the HBM traffic passes)
TAG=$1; PAT=$2; shift; shift
ROOT=$(pwd)
OUT=/tmp/prof_$TAG
rm -rf $OUT; mkdir -p $OUT $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-e2e $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- $BENCH > $OUT/kt.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY -d $OUT/pmc1 -o pmc1 -- $BENCH > $OUT/pmc1.log 2>&1
cd $ROOT
python tools/prof_summary.py $OUT $PAT > gpurun_out/prof_$TAG.json
grep -h '"metric"' $OUT/kt.log > gpurun_out/prof_${TAG}_bench.json
tail -2 $OUT/pmc1.log | cut -c1-300
head -c 3500 gpurun_out/prof_$TAG.json
