#!/bin/bash
# one-pass framing (chained scan) vs the classic three kernels: parity + per-kernel times + the frame workload
T=${1:-r04x}
mkdir -p gpurun_out
export FG_BENCH_CACHE=/tmp/fg_bench_cache
timeout 900 python -m pytest tests -m gpu -x -q -k "fram or utf8 or splitter or chunked or one_pass or frames" > gpurun_out/${T}_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/${T}_pytest.log
bash tools/prof_frame_modes.sh 4 2>&1 | tail -12
cp gpurun_out/frame_modes_kernels.log gpurun_out/${T}_frame_modes_kernels.log
python bench.py --workload frame --steps 5 --warmup 1 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 > gpurun_out/${T}_bench_frame.json
python - <<PY
import json
d=json.loads(open("gpurun_out/${T}_bench_frame.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("frame", round(d["value"]/1e6,1), "M lines/s", round(r["kernel_ms"],3), "ms", d.get("framing"))
PY
