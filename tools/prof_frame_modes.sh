#!/bin/bash
# kernel trace of tools/probe/frame_modes.py: per-kernel times of the one-pass framing scan and of the classic three kernels
ROOT=$(pwd)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fm -o fm -- python $ROOT/tools/probe/frame_modes.py ${1:-4} > /tmp/fm.log 2>&1
grep -E "classic|one-pass|Error|error" /tmp/fm.log | tail -8
cd $ROOT
python - <<PY | tee gpurun_out/frame_modes_kernels.log
import csv,glob
fs=glob.glob("/tmp/fm/**/*kernel_stats.csv", recursive=True)
for f in fs:
    for r in csv.DictReader(open(f)):
        if "frame" in r["Name"] or "fill" in r["Name"].lower() or "memset" in r["Name"].lower(): print(r["Name"][:80], r["Calls"], r["AverageNs"])
PY
