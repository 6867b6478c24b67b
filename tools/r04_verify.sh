#!/bin/bash
# final-tree check: the GPU suite, smoke(), the default bench line as the driver runs it
T=${1:-r04end}
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/${T}_gpu_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${T}_gpu_pytest.log
tail -3 gpurun_out/${T}_gpu_pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py 2> gpurun_out/${T}_bench_default.err | tail -1 > gpurun_out/${T}_bench_default_100M.json
python - <<PY
import json
d=json.loads(open("gpurun_out/${T}_bench_default_100M.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("default", round(d["value"]/1e9,2), "G", round(r["frac"],3), "of copy", round(r.get("frac_of_copy") or 0,3), "traffic/line", round((r.get("traffic") or 0)/d["config"]["lines_per_gpu"],1))
for k in ("configs2","configs3","configs4"):
    c=d[k]; print(k, round(c["value"]/1e6,1), round(c.get("roofline_frac",0),4), {kk:round(vv["lines_per_s"]/1e6,1) for kk,vv in (c.get("e2e") or {}).items() if isinstance(vv,dict) and "lines_per_s" in vv}, c.get("gather_ms"))
print({k: round(v/1e6,1) for k,v in d["e2e"]["aggregate"].items()}, "cpu", round(d["cpu_baseline"]["value"]/1e6,1))
PY
