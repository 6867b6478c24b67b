#!/bin/bash
# round 5, call 3: dispatch policy v2 + launcher-chosen entry reservations: tests, then the same-box A/Bs
T=${1:-r05c}
mkdir -p gpurun_out
export FG_BENCH_CACHE=/tmp/fg_bench_cache
timeout 1200 python -m pytest tests/test_gpu_round5.py -x -q > gpurun_out/${T}_gpu_pytest_round5.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${T}_gpu_pytest_round5.log
tail -5 gpurun_out/${T}_gpu_pytest_round5.log
timeout 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_round5.py > gpurun_out/${T}_gpu_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${T}_gpu_pytest.log
tail -5 gpurun_out/${T}_gpu_pytest.log
export FG_PROBE_SIZES=16384,65536,262144,524288,1048576
FG_PROBE_OPTS=';static_chunks=1' python tools/probe/small_batch.py cfg2 cfg5 ltsv5 cfg4 cfg3 ltsv > gpurun_out/${T}_small_ab.log 2>&1
grep -h "n=" gpurun_out/${T}_small_ab.log
FG_PROBE_SIZES=65536,262144 FG_PROBE_OPTS=';ent_chunk=1;ent_chunk=32;ent_chunk=256;waves_per_cu=8;waves_per_cu=8,ent_chunk=256;waves_per_cu=4' python tools/probe/small_batch.py cfg3 > gpurun_out/${T}_small_gelf_opts.log 2>&1
grep -h "n=" gpurun_out/${T}_small_gelf_opts.log
for w in cfg5 cfg4 ltsv5 cfg3 ltsv; do python tools/sweep.py $w --lines 250000 --reps 16 ';static_chunks=1;;static_chunks=1' 2>&1 | grep "M lines/s"; done | tee gpurun_out/${T}_sweep_4M.log
python tools/sweep.py cfg2 --lines 1000000 --reps 40 ';static_chunks=1;chunk_lines=512;chunk_lines=1024;;static_chunks=1;chunk_lines=512;chunk_lines=1024' 2>&1 | grep "M lines/s" | tee gpurun_out/${T}_sweep_cfg2_40M.log
./tools/probe/store_patterns 4000000 22528 | tee gpurun_out/${T}_store_patterns.log
./tools/probe/store_patterns 4000000 8192 | tee -a gpurun_out/${T}_store_patterns.log
python bench.py --workload cfg5mix --tile-lines 200000 --reps 5 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${T}_bench_cfg5mix_1M.json
python -c "
import json; d=json.loads(open('gpurun_out/${T}_bench_cfg5mix_1M.json').read()); print('cfg5mix 1M', round(d['value']/1e6,1), 'M lines/s', [(s['format'], round(s['lines_per_s']/1e6,1)) for s in d['sub_batches']], 'gather_ms', round(d['gather_ms'],2))"
