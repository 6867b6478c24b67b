#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r02f_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02f_pytest.log
tail -4 gpurun_out/r02f_pytest.log
LS="8 16 32" bash tools/r02_prof_gelf.sh
BENCH_ARGS="--no-cpu-baseline --no-e2e" bash tools/gpu_workloads.sh gpurun_out/r02f_workloads.log cfg4 cfg5 ltsv > /dev/null 2>&1
cat gpurun_out/r02f_workloads.log
