#!/bin/bash
# round 4, GPU call 13: GELF after the 32-bit rank: parity, geometry sweep, phase clocks
T=${1:-r04m}
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q -k "gelf or GELF" > gpurun_out/${T}_pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/${T}_pytest.log
timeout 300 python tools/sweep.py cfg3 --lines 250000 --reps 16 ";lines_per_group=12;lines_per_group=16;lines_per_group=6;gelf_window_kib=4,lines_per_group=10;waves_per_cu=12" 2>/dev/null | grep "M lines/s" | tee gpurun_out/${T}_sweep_cfg3.log
FLOWGGER_AMD_PROF_LIB=1 FG_PROF=1 timeout 200 python tools/sweep.py cfg3 --lines 250000 --reps 4 "" 2>&1 | grep -E "fg prof" | tail -2 | tee gpurun_out/${T}_phases_cfg3.log
