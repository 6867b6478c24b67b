#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r02h_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02h_pytest.log
tail -4 gpurun_out/r02h_pytest.log
BENCH_ARGS="--no-cpu-baseline --no-e2e" bash tools/gpu_workloads.sh gpurun_out/r02h_workloads.log cfg3 cfg4 cfg5 ltsv > /dev/null 2>&1
cat gpurun_out/r02h_workloads.log
