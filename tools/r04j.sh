#!/bin/bash
# round 4, GPU call 10: LTSV head staging (parity, then ltsv5 / cfg5mix), the raw-stream path after the chunk bound
T=${1:-r04j}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_parity.py -m gpu -x -q -k "ltsv or mix or cfg5" > gpurun_out/${T}_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/${T}_pytest.log
timeout 300 python tools/sweep.py ltsv5 --lines 100000 --reps 16 ";no_head=1;tile_cap=12288;tile_cap=24576;tile_cap=32768" 2>/dev/null | grep "M lines/s" | tee gpurun_out/${T}_sweep_ltsv5.log
timeout 300 python bench.py --workload cfg5mix --tile-lines 200000 --reps 10 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${T}_bench_cfg5mix.json 2>/dev/null
python - <<PY
import json
d=json.loads(open("gpurun_out/${T}_bench_cfg5mix.json").read().strip().splitlines()[-1])
print("cfg5mix", round(d["value"]/1e6,1), "M lines/s", [(s["format"], round(s["lines_per_s"]/1e6,1)) for s in d["sub_batches"]], "gather_ms", round(d["gather_ms"],1))
PY
timeout 300 python bench.py --workload cfg4 --tile-lines 250000 --reps 16 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${T}_bench_cfg4.json 2>/dev/null
python - <<PY
import json
d=json.loads(open("gpurun_out/${T}_bench_cfg4.json").read().strip().splitlines()[-1]); e=d.get("e2e",{})
print("cfg4", round(d["value"]/1e6,1), {k: (round(e[k]["lines_per_s"]/1e6,1), round(e[k].get("frac_of_link_h2d") or 0,3)) for k in ("decode_batch","frame_decode_batch") if k in e and "lines_per_s" in e[k]})
PY
