#!/bin/bash
# Builds of this tree against each other on one box (FG_BUILD_VARIANT libraries): tools/lib_ab.sh <tag> "<workloads>" "<libs>" [reps]
# usage (through gpurun): bash tools/lib_ab.sh r05x "cfg4 cfg5" "product libfg_hip_sd3.so" 16,64
tag=${1:-r05x}
wls=${2:-cfg4}
libs=${3:-product}
reps=${4:-16,64}
out=gpurun_out
mkdir -p $out
export FG_BENCH_CACHE=/tmp/fgcache
log=$out/${tag}_lib_ab.log
: > $log
for round in 1 2; do
  for wl in $wls; do
    for lib in $libs; do
      echo "## $lib $wl" >> $log
      if [ "$lib" = product ]; then l=""; else l=$lib; fi
      FLOWGGER_AMD_LIB=$l python tools/sweep.py $wl --lines 250000 --reps $reps '' 2>&1 | grep -v amdgpu.ids >> $log
    done
  done
done
cat $log
