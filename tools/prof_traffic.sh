#!/bin/bash
# usage: tools/prof_traffic.sh <tag> <kernel-substring> [bench args...] -- the two HBM-traffic PMC passes only (FETCH_SIZE,
# WRITE_SIZE in separate runs, MI355X_MICROARCH.md); condensed summary in gpurun_out/traffic_<tag>.json
TAG=$1; PAT=$2; shift; shift
ROOT=$(pwd)
OUT=/tmp/traffic_$TAG
rm -rf $OUT; mkdir -p $OUT $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e $@"
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc3 -o pmc3 -- $BENCH > $OUT/pmc3.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/pmc4 -o pmc4 -- $BENCH > $OUT/pmc4.log 2>&1
cd $ROOT
python tools/prof_summary.py $OUT "$PAT" > gpurun_out/traffic_$TAG.json
grep -h '"metric"' $OUT/pmc3.log | tail -1 > gpurun_out/traffic_${TAG}_bench.json
python -c "
import json; d=json.load(open('gpurun_out/traffic_$TAG.json')); print('$TAG', d.get('hbm_bytes_per_dispatch'), d.get('pmc_per_dispatch_mean'))"
