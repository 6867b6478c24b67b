#!/bin/bash
# kernel iteration: parity of the decode paths + the workloads' rates at 4 M lines
T=${1:-r03k}
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round3.py -m gpu -x -q ${PYTEST_K:+-k "$PYTEST_K"} > gpurun_out/${T}_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${T}_pytest.log
tail -6 gpurun_out/${T}_pytest.log
for w in ${WORKLOADS:-cfg2 cfg4 cfg5 ltsv ltsv5 cfg3}; do python bench.py --workload $w --reps 4 --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-mix 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$w', round(d['value']/1e6,1), 'M lines/s', round(r['kernel_ms'],3), 'ms frac', round(r['frac'],4))"; done
