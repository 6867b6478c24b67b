#!/bin/bash
# after the host-side change of the sliced paths (entry columns per slice): GPU suite, PCIe-inclusive legs of the entry corpora, and the
# headline's HBM traffic again (fg_capi.cpp is part of every workload's source hash)
T=${1:-r03y}
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/${T}_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" >> gpurun_out/${T}_pytest.log
tail -3 gpurun_out/${T}_pytest.log
if [ $rc -ne 0 ]; then tail -60 gpurun_out/${T}_pytest.log | cut -c1-240; exit 1; fi
e2e() { python bench.py --workload $1 --tile-lines 1000000 --reps 4 --steps 3 --warmup 1 --no-cpu-baseline --no-mix 2>/dev/null | tail -1 > gpurun_out/${T}_e2e_$1.json
  python -c "
import json; d=json.loads(open('gpurun_out/${T}_e2e_$1.json').read().strip().splitlines()[-1]); e=d['e2e']; print('$1', {k: round(e[k]['lines_per_s']/1e6,1) for k in e if isinstance(e[k], dict) and 'lines_per_s' in e[k]}, {k: e[k].get('frac_of_link_h2d') for k in e if isinstance(e[k], dict) and 'frac_of_link_h2d' in e[k]})" 2>&1 | tail -1; }
e2e cfg4
bash tools/prof_traffic.sh ${T}_cfg2 k_rfc5424 --reps 40 --no-mix 2>&1 | tail -1
e2e cfg3
e2e ltsv
e2e cfg2
