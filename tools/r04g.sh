#!/bin/bash
T=${1:-r04g}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_round4.py -m gpu -x -q -k "variants or whole_line" > gpurun_out/${T}_pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/${T}_pytest.log
timeout 300 python tools/sweep.py cfg4 --lines 250000 --reps 16 ";sd_walk=1;tile_cap=10240;tile_cap=14336;tile_cap=16384" 2>/dev/null | grep "M lines/s" | tee gpurun_out/${T}_sweep_cfg4.log
timeout 300 python tools/sweep.py cfg5 --lines 100000 --reps 16 ";sd_walk=1;tile_cap=12288" 2>/dev/null | grep "M lines/s" | tee gpurun_out/${T}_sweep_cfg5.log
FLOWGGER_AMD_PROF_LIB=1 FG_PROF=1 timeout 200 python tools/sweep.py cfg4 --lines 250000 --reps 4 "" 2>&1 | grep -E "fg prof" | tail -2 | tee gpurun_out/${T}_phases_cfg4.log
