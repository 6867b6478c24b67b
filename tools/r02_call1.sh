#!/bin/bash
# round 2, GPU call 1: the whole -m gpu suite, the default bench line (self-launch path too), cfg2 PMC passes
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r02a_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02a_pytest.log
tail -3 gpurun_out/r02a_pytest.log
python bench.py > gpurun_out/r02a_bench_default.json 2> gpurun_out/r02a_bench_default.err; echo "bench rc=$?"
tail -c 2500 gpurun_out/r02a_bench_default.json
python bench.py --workload cfg1 --reps 4 --steps 5 --warmup 2 > gpurun_out/r02a_bench_cfg1.json 2> gpurun_out/r02a_bench_cfg1.err; echo "cfg1 rc=$?"
bash tools/prof.sh r02a_cfg2 > gpurun_out/r02a_prof_cfg2.log 2>&1
tail -5 gpurun_out/r02a_prof_cfg2.log
BENCH_ARGS="--no-cpu-baseline --no-e2e" bash tools/gpu_workloads.sh gpurun_out/r02a_workloads.log cfg4 cfg5 cfg3 ltsv rfc3164 > /dev/null 2>&1
cat gpurun_out/r02a_workloads.log
