#!/bin/bash
# the tree after the raw-stream path kept per-slice entries for RFC5424 only: the headline's HBM traffic under this source hash, the
# GELF / LTSV raw-stream tests, their PCIe-inclusive legs
T=${1:-r03x}
mkdir -p gpurun_out
bash tools/prof_traffic.sh ${T}_cfg2 k_rfc5424 --reps 40 --no-mix 2>&1 | tail -1
python -m pytest tests/test_gpu_round3.py -m gpu -x -q -k "raw_stream_pipelined and (gelf or ltsv)" > gpurun_out/${T}_pytest_frame.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${T}_pytest_frame.log
tail -2 gpurun_out/${T}_pytest_frame.log
for w in cfg3 ltsv; do python bench.py --workload $w --tile-lines 1000000 --reps 4 --steps 3 --warmup 1 --no-cpu-baseline --no-mix 2>/dev/null | tail -1 > gpurun_out/${T}_e2e_$w.json
  python -c "
import json; d=json.loads(open('gpurun_out/${T}_e2e_$w.json').read().strip().splitlines()[-1]); e=d['e2e']; print('$w', {k: round(e[k]['lines_per_s']/1e6,1) for k in e if isinstance(e[k], dict) and 'lines_per_s' in e[k]})" 2>&1 | tail -1; done
