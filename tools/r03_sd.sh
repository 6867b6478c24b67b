#!/bin/bash
T=${1:-r03sd}
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rfc5424 or cfg4 or cfg2 or sd or fuzz or variants or long or overflow or vectors or error_table or edge or replicas or cfg5" > gpurun_out/${T}_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${T}_pytest.log
tail -6 gpurun_out/${T}_pytest.log
for w in cfg2 cfg4 cfg5; do python bench.py --workload $w --reps 4 --steps 5 --warmup 2 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$w', round(d['value']/1e6,1), 'M lines/s', round(r['kernel_ms'],3), 'ms frac', round(r['frac'],4))"; done
