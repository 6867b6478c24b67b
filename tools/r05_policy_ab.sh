#!/bin/bash
# The launch policies of this tree against libfg_hip_r05z.so (the tree of the round's first closing run) at 0.5 M .. 16 M lines per
# launch, alternated on one box.  usage (through gpurun): bash tools/r05_policy_ab.sh <tag>
tag=${1:-r05v}
out=gpurun_out
mkdir -p $out
export FG_BENCH_CACHE=/tmp/fgcache
log=$out/${tag}_policy_ab.log
: > $log
run() { echo "## ${FLOWGGER_AMD_LIB:-product} $*" >> $log; "$@" 2>&1 | grep -v amdgpu.ids >> $log; }
for round in 1 2; do
  for wl in cfg3 cfg4 cfg5 ltsv; do
    run python tools/sweep.py $wl --lines 250000 --reps 2,4,8,16,64 ';'
    FLOWGGER_AMD_LIB=libfg_hip_r05z.so run python tools/sweep.py $wl --lines 250000 --reps 2,4,8,16,64 ';'
  done
done
cat $log
