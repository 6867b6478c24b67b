#!/bin/bash
# round 5, call 4: fused one-launch encoder + 16-byte stores, stage-A classification of the pair-parallel kernel (A/B against a variant
# build of the same tree), tickets drawn at chunk entry, entry reservations >= 256 slots
T=${1:-r05d}
mkdir -p gpurun_out
export FG_BENCH_CACHE=/tmp/fg_bench_cache
timeout 1500 python -m pytest tests/test_gpu_round5.py -x -q > gpurun_out/${T}_gpu_pytest_round5.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${T}_gpu_pytest_round5.log
tail -5 gpurun_out/${T}_gpu_pytest_round5.log
timeout 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_round5.py > gpurun_out/${T}_gpu_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${T}_gpu_pytest.log
tail -5 gpurun_out/${T}_gpu_pytest.log
# the encoder: one launch against three, same box
for o in "" "--launch-opts encode_three_pass=1"; do
  python bench.py --workload cfg1 --reps 4 --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-calib $o 2>/dev/null | tail -1 > gpurun_out/${T}_cfg1_tmp.json
  python -c "
import json; d=json.loads(open('gpurun_out/${T}_cfg1_tmp.json').read()); e=d['encode']; print('cfg1 [$o] encode ms', round(e['ms'],3), 'lines/s', round(e['lines_per_s']/1e6,1), 'M; decode ms', round(d['roofline']['kernel_ms'],3))"
  cp gpurun_out/${T}_cfg1_tmp.json "gpurun_out/${T}_bench_cfg1$(echo $o | tr -d ' =-').json"
done
# stage-A classification: product library against the variant without it (same tree otherwise), alternated
for r in 1 2; do
  python tools/sweep.py cfg4 --lines 250000 --reps 16 ';' 2>&1 | grep "M lines/s" | sed 's/^/stageA    /'
  FLOWGGER_AMD_LIB=libfg_hip_nostagea.so python tools/sweep.py cfg4 --lines 250000 --reps 16 ';' 2>&1 | grep "M lines/s" | sed 's/^/no-stageA /'
done | tee gpurun_out/${T}_ab_stagea_cfg4.log
export FG_PROBE_SIZES=16384,65536,262144,524288,1048576
FG_PROBE_OPTS=';static_chunks=1' python tools/probe/small_batch.py cfg2 cfg5 cfg4 cfg3 > gpurun_out/${T}_small_ab.log 2>&1
grep -h "n=" gpurun_out/${T}_small_ab.log
FG_PROBE_SIZES=1048576,2097152,4194304 FG_PROBE_OPTS=';static_chunks=1' python tools/probe/small_batch.py cfg2 > gpurun_out/${T}_small_cfg2_big.log 2>&1
grep -h "n=" gpurun_out/${T}_small_cfg2_big.log
python tools/sweep.py cfg2 --lines 1000000 --reps 40 ';static_chunks=1;;static_chunks=1' 2>&1 | grep "M lines/s" | tee gpurun_out/${T}_sweep_cfg2_40M.log
