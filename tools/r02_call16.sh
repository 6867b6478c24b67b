#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r02m_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02m_pytest.log
tail -3 gpurun_out/r02m_pytest.log
bash tools/prof.sh r02m_cfg2 --reps 40 > gpurun_out/r02m_prof_cfg2.log 2>&1
PAT=k_gelf bash tools/prof.sh r02m_cfg3 --workload cfg3 --tile-lines 1000000 --reps 4 > gpurun_out/r02m_prof_cfg3.log 2>&1
BENCH_ARGS="--no-cpu-baseline --no-e2e" bash tools/gpu_workloads.sh gpurun_out/r02m_workloads.log cfg3 cfg4 cfg5 ltsv rfc3164 > /dev/null 2>&1
cat gpurun_out/r02m_workloads.log
