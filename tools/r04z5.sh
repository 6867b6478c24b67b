#!/bin/bash
T=${1:-r04z5}
mkdir -p gpurun_out
export FG_BENCH_CACHE=/tmp/fg_bench_cache
timeout 400 python tools/sweep.py cfg4 --lines 250000 --reps 500 ";chunk_lines=2048;;chunk_lines=2048;chunk_lines=4096;chunk_lines=1536" 2>/dev/null | grep "M lines/s" | tee gpurun_out/${T}_sweep_cfg4_125M.log
timeout 400 python tools/sweep.py cfg5 --lines 250000 --reps 16 ";chunk_lines=2048;;chunk_lines=512;;chunk_lines=4096" 2>/dev/null | grep "M lines/s" | tee gpurun_out/${T}_sweep_cfg5.log
