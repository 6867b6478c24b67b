#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r02ab.log; : > $O
python -m pytest tests -m gpu -x -q -k "encode or transcode or pipeline or Encoder or merger" > gpurun_out/r02ab_pytest.log 2>&1; echo "pytest rc=$?" >> $O
tail -2 gpurun_out/r02ab_pytest.log >> $O
python bench.py --workload cfg1 --reps 4 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r02ab_cfg1.json
python -c "
import json; d=json.loads(open('gpurun_out/r02ab_cfg1.json').read().strip().splitlines()[-1]); print('cfg1', d['value']/1e6, d.get('encode',{}), json.dumps(d.get('e2e'))[:700])" >> $O
cat $O
