#!/bin/bash
# the last call of the round: GPU suite + smoke + the default bench line on the final tree; LTSV (kernel sources changed since the closing
# run): full-size bench, HBM traffic for the restamp
T=${1:-r04last}
mkdir -p gpurun_out
export FG_BENCH_CACHE=/tmp/fg_bench_cache
python -m pytest tests -m gpu -x -q > gpurun_out/${T}_gpu_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${T}_gpu_pytest.log
tail -3 gpurun_out/${T}_gpu_pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
env -u FG_BENCH_CACHE python bench.py 2> gpurun_out/${T}_bench_default.err | tail -1 > gpurun_out/${T}_bench_default_100M.json
cut -c1-220 gpurun_out/${T}_bench_default_100M.json
bash tools/prof_traffic.sh ${T}_ltsv k_ltsv --workload ltsv --tile-lines 250000 --reps 16 --no-calib > /dev/null 2>&1
bash tools/prof_traffic.sh ${T}_ltsv5 k_ltsv --workload ltsv5 --tile-lines 250000 --reps 16 --no-calib > /dev/null 2>&1
PAT=k_ltsv bash tools/prof_quick.sh ${T}_ltsv k_ltsv --workload ltsv --tile-lines 250000 --reps 16 --no-calib > /dev/null 2>&1
python bench.py --workload ltsv --tile-lines 250000 --reps 400 --steps 10 --warmup 2 --no-e2e --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${T}_bench_ltsv_100M.json
python bench.py --workload ltsv5 --tile-lines 250000 --reps 80 --steps 5 --warmup 1 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 > gpurun_out/${T}_bench_ltsv5_20M.json
python - <<PY
import json
def ld(f): return json.loads(open(f"gpurun_out/${T}_"+f+".json").read().strip().splitlines()[-1])
d=ld("bench_default_100M"); r=d["roofline"]
print("default", round(d["value"]/1e9,2), "G", round(r["frac"],3), "of copy", round(r.get("frac_of_copy") or 0,3), "traffic", r.get("traffic"))
for k in ("configs2","configs3","configs4"):
    c=d[k]; print(k, round(c["value"]/1e6,1), round(c.get("roofline_frac",0),4), {kk:round(vv["lines_per_s"]/1e6,1) for kk,vv in (c.get("e2e") or {}).items() if isinstance(vv,dict) and "lines_per_s" in vv}, c.get("gather_ms"))
print({k: round(v/1e6,1) for k,v in d["e2e"]["aggregate"].items()})
for f in ("bench_ltsv_100M","bench_ltsv5_20M"):
    d=ld(f); r=d["roofline"]; print(f, round(d["value"]/1e6,1), round(r["kernel_ms"],2), round(r["frac"],4))
p=json.load(open("gpurun_out/prof_${T}_ltsv.json")); print("ltsv dispatch", p["dispatch_info"])
PY
