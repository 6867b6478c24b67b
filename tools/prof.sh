#!/bin/bash
# usage: tools/prof.sh <tag> [bench args...]  -- kernel trace + PMC passes of bench.py under rocprofv3;
# leaves only a condensed summary (gpurun_out/prof_<tag>.json) -- raw traces are deleted.
TAG=$1; shift
ROOT=$(pwd)
OUT=/tmp/prof_$TAG
rm -rf $OUT; mkdir -p $OUT $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-e2e $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- $BENCH > $OUT/kt.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $OUT/pmc1 -o pmc1 -- $BENCH > $OUT/pmc1.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD -d $OUT/pmc2 -o pmc2 -- $BENCH > $OUT/pmc2.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc3 -o pmc3 -- $BENCH > $OUT/pmc3.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/pmc4 -o pmc4 -- $BENCH > $OUT/pmc4.log 2>&1
cd $ROOT
python tools/prof_summary.py $OUT "${PAT:-k_}" > gpurun_out/prof_$TAG.json
grep -h '"metric"' $OUT/kt.log > gpurun_out/prof_${TAG}_bench.json
tail -2 $OUT/pmc1.log | cut -c1-300
cat gpurun_out/prof_$TAG.json | head -80
