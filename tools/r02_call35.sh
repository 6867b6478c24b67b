#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r02ae.log; : > $O
python -m pytest tests -m gpu -x -q -k "gelf or Gelf or GELF or framed or threads or sharded" > gpurun_out/r02ae_pytest.log 2>&1; echo "pytest rc=$?" >> $O
tail -2 gpurun_out/r02ae_pytest.log >> $O
run() { w=$1; shift; echo "== $w $*" >> $O; env "$@" python bench.py --workload $w --tile-lines 1000000 --reps 4 --steps 5 --warmup 1 --no-cpu-baseline --no-e2e 2> gpurun_out/err.txt | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value']/1e6, d['roofline'].get('kernel_ms'), d['roofline'].get('frac'))" >> $O; }
run cfg3 A=1
python bench.py --workload cfg3 --steps 10 --warmup 2 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cfg3 100M', d['value']/1e6, d['roofline'].get('kernel_ms'), d['roofline'].get('frac'))" >> $O
cat $O
