#!/bin/bash
# k_gelf_general with its pending lines staged in LDS (this tree) against the walk through global memory (libfg_hip_enca.so), one box.
# usage (through gpurun): bash tools/r05_gelf_general_ab.sh <tag>
tag=${1:-r05aa}
out=gpurun_out
mkdir -p $out
export FG_BENCH_CACHE=/tmp/fgcache
python -m pytest tests -x -q -m gpu -k "gelf" > $out/${tag}_gpu_pytest_gelf.log 2>&1
echo "pytest rc=$?" >> $out/${tag}_gpu_pytest_gelf.log
tail -3 $out/${tag}_gpu_pytest_gelf.log
log=$out/${tag}_gelf_general_ab.log
: > $log
for round in 1 2; do
  for lib in product libfg_hip_enca.so; do
    if [ "$lib" = product ]; then l=""; else l=$lib; fi
    echo "## $lib" >> $log
    FG_PROBE_SIZES=4096,16384,65536,262144 FG_PROBE_TOP=262144 FLOWGGER_AMD_LIB=$l python tools/probe/small_batch.py cfg3 2>&1 | grep "n=" >> $log
    FLOWGGER_AMD_LIB=$l python tools/sweep.py cfg3 --lines 250000 --reps 4,16,64 '' 2>&1 | grep "n=" >> $log
  done
done
cat $log
