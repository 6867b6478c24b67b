#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r02t.log; : > $O
python -m pytest tests -m gpu -x -q -k "gelf or Gelf or GELF" > gpurun_out/r02t_pytest.log 2>&1; echo "pytest rc=$?" >> $O
tail -2 gpurun_out/r02t_pytest.log >> $O
run() { echo "== $*" >> $O; env "$@" FG_PLAN=1 python bench.py --workload cfg3 --steps 10 --warmup 2 --no-cpu-baseline --no-e2e 2> gpurun_out/err.txt | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value']/1e6, d['roofline'].get('kernel_ms'), d['roofline'].get('frac'))" >> $O; grep -m1 "gelf plan" gpurun_out/err.txt >> $O; }
run A=1
FG_PROF=1 python bench.py --workload cfg3 --steps 1 --warmup 1 --no-cpu-baseline --no-e2e 2>&1 | grep -A1 -m1 "prof" >> $O
cat $O
