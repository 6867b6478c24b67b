#!/bin/bash
export FG_BENCH_CACHE=/tmp/fg_bench_cache
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q -k "ltsv or LTSV or cfg5mix or mixed" > gpurun_out/r04t2_pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r04t2_pytest.log
timeout 300 python tools/sweep.py ltsv5 --lines 250000 --reps 16 ";;" 2>/dev/null | grep "M lines/s" | tee gpurun_out/r04t2_sweep_ltsv5.log
python bench.py --workload ltsv5 --tile-lines 250000 --reps 80 --steps 5 --warmup 1 --no-cpu-baseline --no-e2e --no-calib 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ltsv5 20M', round(d['value']/1e6,1), round(d['roofline']['kernel_ms'],2))"
python bench.py --workload cfg5mix --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg5mix', round(d['value']/1e6,1), d['gather_ms'], [ (s['format'], round(s['kernel_ms'],2)) for s in d.get('sub_batches',[])])"
