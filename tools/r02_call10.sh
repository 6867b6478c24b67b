#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -k "gelf or Gelf or entry_workloads" > gpurun_out/r02i_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02i_pytest.log
tail -3 gpurun_out/r02i_pytest.log
LS="16" bash tools/r02_prof_gelf.sh
echo "## cfg2 tile/waves experiments" > gpurun_out/r02i_cfg2.log
for envs in "X=1" "FG_TILE_CAP=17408" "FG_TILE_CAP=17408 FG_WAVES_PER_CU=8" "FG_LINES_PER_GROUP=32"; do
  echo "## $envs" >> gpurun_out/r02i_cfg2.log
  env $envs python bench.py --steps 10 --warmup 2 --reps 40 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); r=d['roofline']; print(json.dumps({'glines_s':round(d['value']/1e9,3),'kernel_ms':round(r['kernel_ms'],4),'frac':round(r['frac'],4)}))" >> gpurun_out/r02i_cfg2.log
done
cat gpurun_out/r02i_cfg2.log
