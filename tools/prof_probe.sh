#!/bin/bash
# kernel trace of a probe script: tools/prof_probe.sh <script> [args...]; prints per-kernel calls and average durations
ROOT=$(pwd)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o pp -- python $ROOT/$1 "${@:2}" > /tmp/pp.log 2>&1
grep -E "lines|Error|error" /tmp/pp.log | tail -12
cd $ROOT
python - <<PY | tee gpurun_out/prof_probe_kernels.log
import csv,glob
for f in glob.glob("/tmp/pp/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(r["Name"][:70], r["Calls"], round(float(r["AverageNs"])/1e3,1), "us avg", round(float(r["TotalDurationNs"])/1e6,2), "ms total")
PY
