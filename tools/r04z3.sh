#!/bin/bash
T=${1:-r04z3}
mkdir -p gpurun_out
export FG_BENCH_CACHE=/tmp/fg_bench_cache
timeout 300 python tools/sweep.py cfg2 --lines 1000000 --reps 8 ";chunk_lines=128;;chunk_lines=192;;chunk_lines=320;;chunk_lines=384;;chunk_lines=512;;chunk_lines=64" 2>/dev/null | grep "M lines/s" | tee gpurun_out/${T}_sweep_cfg2.log
timeout 300 python tools/sweep.py ltsv --lines 250000 --reps 16 ";chunk_lines=128;;chunk_lines=384;;chunk_lines=512;;chunk_lines=640" 2>/dev/null | grep "M lines/s" | tee gpurun_out/${T}_sweep_ltsv.log
timeout 300 python tools/sweep.py cfg4 --lines 250000 --reps 16 ";chunk_lines=512;;chunk_lines=768;;chunk_lines=1536;;chunk_lines=2048" 2>/dev/null | grep "M lines/s" | tee gpurun_out/${T}_sweep_cfg4.log
python bench.py --workload cfg3 --tile-lines 250000 --reps 400 --steps 10 --warmup 2 --no-e2e --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${T}_bench_cfg3_100M.json
python - <<PY
import json
d=json.loads(open("gpurun_out/${T}_bench_cfg3_100M.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("cfg3 100M", round(d["value"]/1e6,1), "M lines/s", round(r["kernel_ms"],3), "ms frac", round(r["frac"],4))
PY
