#!/bin/bash
# GELF: five waves per SIMD (96 registers, 8 040 B of LDS: rare paths by shuffle instead of LDS key arrays)
T=${1:-r04s}
mkdir -p gpurun_out
export FG_BENCH_CACHE=/tmp/fg_bench_cache
timeout 600 python -m pytest tests -m gpu -x -q -k "gelf or GELF or cfg3" > gpurun_out/${T}_pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/${T}_pytest.log
timeout 300 python tools/sweep.py cfg3 --lines 250000 --reps 16 ";waves_per_cu=16;waves_per_cu=18;waves_per_cu=20;chunk_lines=512;chunk_lines=128" 2>/dev/null | grep "M lines/s" | tee gpurun_out/${T}_sweep_cfg3.log
python bench.py --workload cfg3 --tile-lines 250000 --reps 400 --steps 10 --warmup 2 --no-e2e --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${T}_bench_cfg3_100M.json
python - <<PY
import json
d=json.loads(open("gpurun_out/${T}_bench_cfg3_100M.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("cfg3 100M", round(d["value"]/1e6,1), "M lines/s", round(r["kernel_ms"],3), "ms frac", round(r["frac"],4))
PY
