#!/bin/bash
# round 4, GPU call 4: pair-parallel SD walk v2 (slot table built once, no barrier inside the slot loops): parity, sweep, phase clocks
T=${1:-r04d}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_round4.py -m gpu -x -q -k "variants or whole_line" > gpurun_out/${T}_pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/${T}_pytest.log
timeout 300 python tools/sweep.py cfg4 --lines 250000 --reps 16 ";sd_walk=1;tile_cap=8192;tile_cap=10240;tile_cap=14336;tile_cap=16384;tile_cap=12288,chunk_lines=1024;tile_cap=12288,chunk_lines=256" 2>/dev/null | grep "M lines/s" | tee gpurun_out/${T}_sweep_cfg4.log
timeout 300 python tools/sweep.py cfg5 --lines 100000 --reps 16 ";sd_walk=1;tile_cap=16384;tile_cap=19456" 2>/dev/null | grep "M lines/s" | tee gpurun_out/${T}_sweep_cfg5.log
FLOWGGER_AMD_PROF_LIB=1 FG_PROF=1 timeout 300 python tools/sweep.py cfg4 --lines 250000 --reps 4 "" 2>&1 | grep "fg prof" | tail -4 | tee gpurun_out/${T}_phases_cfg4.log
