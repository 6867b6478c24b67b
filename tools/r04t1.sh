#!/bin/bash
export FG_BENCH_CACHE=/tmp/fg_bench_cache
mkdir -p gpurun_out
timeout 300 python tools/sweep.py ltsv5 --lines 250000 --reps 16 ";tile_cap=24576;tile_cap=28672;tile_cap=36864;tile_cap=14336;;tile_cap=24576" 2>/dev/null | grep "M lines/s" | tee gpurun_out/r04t1_sweep_ltsv5.log
timeout 300 python tools/sweep.py cfg5 --lines 250000 --reps 16 ";tile_cap=16384;tile_cap=20480;tile_cap=24576;tile_cap=8192;" 2>/dev/null | grep "M lines/s" | tee gpurun_out/r04t1_sweep_cfg5.log
