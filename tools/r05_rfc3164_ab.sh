#!/bin/bash
# RFC3164 on the shared streaming pipeline (this tree) against the one-workgroup-per-64-lines kernel of rounds 1-4
# (libfg_hip_r05z.so), one box.  usage (through gpurun): bash tools/r05_rfc3164_ab.sh <tag>
tag=${1:-r05ab}
out=gpurun_out
mkdir -p $out
export FG_BENCH_CACHE=/tmp/fgcache
python -m pytest tests -x -q -m gpu -k "3164" > $out/${tag}_gpu_pytest_rfc3164.log 2>&1
echo "pytest rc=$?" >> $out/${tag}_gpu_pytest_rfc3164.log
tail -3 $out/${tag}_gpu_pytest_rfc3164.log
log=$out/${tag}_rfc3164_ab.log
: > $log
for round in 1 2; do
  for lib in product libfg_hip_r05z.so; do
    if [ "$lib" = product ]; then l=""; else l=$lib; fi
    echo "## $lib" >> $log
    FLOWGGER_AMD_LIB=$l python bench.py --workload rfc3164 --tile-lines 250000 --reps 400 --steps 10 --warmup 2 --no-e2e --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('rfc3164 100 M lines', round(d['value']/1e6,1), 'M lines/s', round(r['kernel_ms'],3), 'ms frac', round(r['frac'],4))" >> $log
    FG_PROBE_SIZES=16384,65536,262144,1048576 FG_PROBE_TOP=1048576 FLOWGGER_AMD_LIB=$l python tools/probe/small_batch.py rfc3164 2>&1 | grep "n=" >> $log
  done
done
cat $log
