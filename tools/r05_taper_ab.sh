#!/bin/bash
# Same-box sweep of the ticket dispatch's chunk size and taper depth (fg_plan_policy.hpp) per format and batch size.
# usage (through gpurun): bash tools/r05_taper_ab.sh <tag>
tag=${1:-r05t}
out=gpurun_out
mkdir -p $out
export FG_BENCH_CACHE=/tmp/fgcache
log=$out/${tag}_taper_ab.log
: > $log
run() { echo "## $*" >> $log; "$@" >> $log 2>&1; }
T0=no_taper=1
run python tools/sweep.py cfg3 --lines 250000 --reps 16 ";chunk_lines=64,$T0;chunk_lines=64,taper_levels=1;chunk_lines=64,taper_levels=2;chunk_lines=64;chunk_lines=96,$T0;chunk_lines=48,$T0;chunk_lines=128,$T0;chunk_lines=64,$T0"
run python tools/sweep.py cfg3 --lines 250000 --reps 64 ";chunk_lines=64,$T0;chunk_lines=64,taper_levels=1;chunk_lines=128,$T0;chunk_lines=256,$T0;"
run python tools/sweep.py cfg4 --lines 250000 --reps 16 ";chunk_lines=64,$T0;chunk_lines=64;chunk_lines=128,$T0;chunk_lines=128,taper_levels=1;chunk_lines=128,taper_levels=2;chunk_lines=128;chunk_lines=256;chunk_lines=512;chunk_lines=96;chunk_lines=192;"
run python tools/sweep.py cfg4 --lines 250000 --reps 64 ";chunk_lines=128,$T0;chunk_lines=128;chunk_lines=256,$T0;chunk_lines=256;chunk_lines=512,$T0;chunk_lines=512;chunk_lines=1024,$T0;chunk_lines=1024;"
run python tools/sweep.py ltsv --lines 250000 --reps 16 ";chunk_lines=128,$T0;chunk_lines=128,taper_levels=1;chunk_lines=256,$T0;chunk_lines=256,taper_levels=1;chunk_lines=256,taper_levels=2;"
run python tools/sweep.py ltsv --lines 250000 --reps 64 ";chunk_lines=128,$T0;chunk_lines=128,taper_levels=1;chunk_lines=256,$T0;chunk_lines=256,taper_levels=2;"
run python tools/sweep.py cfg2 --lines 1000000 --reps 40 ";$T0;taper_levels=1;$T0;taper_levels=1"
run python tools/sweep.py cfg5 --lines 250000 --reps 16 ";chunk_lines=128,$T0;chunk_lines=128;chunk_lines=256,$T0;chunk_lines=256;chunk_lines=512;chunk_lines=64;"
run python tools/sweep.py ltsv5 --lines 250000 --reps 16 ";chunk_lines=128,$T0;chunk_lines=128;chunk_lines=256,$T0;chunk_lines=256;chunk_lines=512;chunk_lines=64;"
cat $log
