#!/bin/bash
# The last GPU call of round 6, on the final kernel sources: the GPU suite, smoke, and the PMC passes profiles/traffic.json is stamped
# from (tools/update_traffic.py, run afterwards where the repository is): kernel trace + instruction counters + traffic for the
# headline, GELF and structured data; traffic for the others.
T=${1:-r06end}
mkdir -p gpurun_out
export FG_BENCH_CACHE=/tmp/fg_bench_cache
python -m pytest tests -m gpu -x -q > gpurun_out/${T}_gpu_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${T}_gpu_pytest.log
tail -3 gpurun_out/${T}_gpu_pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
PAT=k_rfc5424 bash tools/prof.sh ${T}_cfg2 --reps 100 --no-mix --no-legs --no-calib > gpurun_out/${T}_prof_cfg2.log 2>&1
PAT='k_gelf<' bash tools/prof.sh ${T}_cfg3 --workload cfg3 --tile-lines 250000 --reps 16 --no-calib > gpurun_out/${T}_prof_cfg3.log 2>&1
PAT=k_rfc5424 bash tools/prof.sh ${T}_cfg4 --workload cfg4 --tile-lines 250000 --reps 16 --no-calib > gpurun_out/${T}_prof_cfg4.log 2>&1
bash tools/prof_traffic.sh ${T}_cfg5 k_rfc5424 --workload cfg5 --tile-lines 250000 --reps 16 --no-calib > /dev/null 2>&1
bash tools/prof_traffic.sh ${T}_ltsv k_ltsv --workload ltsv --tile-lines 250000 --reps 16 --no-calib > /dev/null 2>&1
bash tools/prof_traffic.sh ${T}_ltsv5 k_ltsv --workload ltsv5 --tile-lines 250000 --reps 16 --no-calib > /dev/null 2>&1
bash tools/prof_traffic.sh ${T}_rfc3164 k_rfc3164 --workload rfc3164 --tile-lines 250000 --reps 16 --no-calib > /dev/null 2>&1
env -u FG_BENCH_CACHE python bench.py 2> gpurun_out/${T}_bench_default.err | tail -1 > gpurun_out/${T}_bench_default_100M.json
cut -c1-300 gpurun_out/${T}_bench_default_100M.json
ls gpurun_out | grep -c $T
