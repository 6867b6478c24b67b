#!/bin/bash
# round 4, GPU call 5: where does the pair-parallel kernel wait?  ablations of the measurement build (FG_ABLATE: 1 = no row stores,
# 4 = no entry stores, 2 = no stage B) against each other, and the waves-per-CU scaling of the product kernel
T=${1:-r04e}
mkdir -p gpurun_out
for ab in 0 1 4 5 2; do
  echo "== FG_ABLATE=$ab"
  FLOWGGER_AMD_PROF_LIB=1 FG_PROF=1 FG_ABLATE=$ab timeout 200 python tools/sweep.py cfg4 --lines 250000 --reps 4 "" 2>&1 | grep -E "fg prof|M lines" | tail -3
done 2>&1 | tee gpurun_out/${T}_ablate_cfg4.log
timeout 300 python tools/sweep.py cfg4 --lines 250000 --reps 16 "chunk_lines=1024;chunk_lines=1024,waves_per_cu=7;chunk_lines=1024,waves_per_cu=6;chunk_lines=1024,waves_per_cu=4;chunk_lines=2048;chunk_lines=512" 2>/dev/null | grep "M lines/s" | tee gpurun_out/${T}_sweep_cfg4.log
