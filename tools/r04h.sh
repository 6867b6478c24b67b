#!/bin/bash
# round 4, GPU call 8: the whole GPU suite on the new kernels, then the default bench line and the SD / long-tail workloads at size
T=${1:-r04h}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_gpu_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/${T}_gpu_pytest.log
timeout 400 python bench.py > gpurun_out/${T}_bench_default.json 2> gpurun_out/${T}_bench_default.err; echo "bench rc=$?"; tail -c 300 gpurun_out/${T}_bench_default.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04h_bench_default.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("cfg2", d["value"], r["frac"], r.get("copy_GBps"), r.get("frac_of_copy"))
print({k: (round(d[k].get("value",0)/1e6,1), round(d[k].get("roofline_frac",0),4), d[k].get("error")) for k in ("configs2","configs3","configs4") if k in d})
print({k: round(v/1e6,1) for k,v in d["e2e"]["aggregate"].items()})
PY
timeout 300 python tools/sweep.py cfg5 --lines 100000 --reps 16 ";tile_cap=8192;tile_cap=10240;tile_cap=14336;sd_walk=1" 2>/dev/null | grep "M lines/s" | tee gpurun_out/${T}_sweep_cfg5.log
timeout 300 python tools/sweep.py cfg4 --lines 250000 --reps 16 ";tile_cap=11264;tile_cap=13312;chunk_lines=2048" 2>/dev/null | grep "M lines/s" | tee gpurun_out/${T}_sweep_cfg4.log
