#!/bin/bash
# after a change of the kernel sources: the default bench line and the PMC / traffic profiles profiles/traffic.json is keyed on
T=${1:-r02z}
mkdir -p gpurun_out
python bench.py 2> gpurun_out/${T}_bench_default.err | tail -1 > gpurun_out/${T}_bench_default_100M.json
cut -c1-260 gpurun_out/${T}_bench_default_100M.json
bash tools/prof.sh ${T}_cfg2 --reps 40 > gpurun_out/${T}_prof_cfg2.log 2>&1
PAT='k_gelf<' bash tools/prof.sh ${T}_cfg3 --workload cfg3 --tile-lines 1000000 --reps 4 > gpurun_out/${T}_prof_cfg3.log 2>&1
bash tools/prof_traffic.sh ${T}_ltsv k_ltsv --workload ltsv --tile-lines 1000000 --reps 4 > /dev/null 2>&1
bash tools/prof_traffic.sh ${T}_cfg4 k_rfc5424 --workload cfg4 --tile-lines 1000000 --reps 4 > /dev/null 2>&1
python bench.py --workload cfg4 --tile-lines 200000 --reps 20 --steps 5 --warmup 1 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 | cut -c80-200
ls gpurun_out | grep -c $T
