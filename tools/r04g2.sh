#!/bin/bash
T=${1:-r04g2}
mkdir -p gpurun_out
export FG_BENCH_CACHE=/tmp/fg_bench_cache
timeout 600 python -m pytest tests -m gpu -x -q -k "gelf or GELF or cfg3 or fram or chunked or splitter" > gpurun_out/${T}_pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/${T}_pytest.log
bash tools/prof_probe.sh tools/probe/frame_sizes.py cfg3 2>&1 | tail -14
cp gpurun_out/prof_probe_kernels.log gpurun_out/${T}_probe_frame_gelf.log
python bench.py --workload cfg3 --tile-lines 250000 --reps 400 --steps 10 --warmup 2 --no-e2e --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${T}_bench_cfg3_100M.json
bash tools/prof_traffic.sh ${T}_cfg3 'k_gelf<' --workload cfg3 --tile-lines 250000 --reps 16 --no-calib > /dev/null 2>&1
python - <<PY
import json
d=json.loads(open("gpurun_out/${T}_bench_cfg3_100M.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("cfg3 100M", round(d["value"]/1e6,1), "M lines/s", round(r["kernel_ms"],3), "ms frac", round(r["frac"],4))
PY
