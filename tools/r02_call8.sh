#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r02g_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02g_pytest.log
tail -4 gpurun_out/r02g_pytest.log
LS="12 16 24" bash tools/r02_prof_gelf.sh
echo "## W3" >> gpurun_out/r02d_cfg3.log
for L in 16 24 32; do
FG_GELF_W3=1 FG_LINES_PER_GROUP=$L python bench.py --workload cfg3 --tile-lines 200000 --reps 20 --steps 5 --warmup 1 --no-cpu-baseline --no-e2e 2>> gpurun_out/r02d_cfg3.err | tail -1 | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); r=d['roofline']; print(json.dumps({'Mlines_s':round(d['value']/1e6,1),'kernel_ms':round(r['kernel_ms'],3),'frac':round(r['frac'],4)}))" >> gpurun_out/r02d_cfg3.log
done
tail -4 gpurun_out/r02d_cfg3.log
