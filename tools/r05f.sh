#!/bin/bash
# round 5, call 6: dense device merge + threads launcher tests; what the encoder's write kernel waits for (PMC); one chunk per wave
# below the ticket threshold
T=${1:-r05f}
mkdir -p gpurun_out
export FG_BENCH_CACHE=/tmp/fg_bench_cache
timeout 1200 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round4.py -x -q -k "merge or threads or zero_copy" > gpurun_out/${T}_gpu_pytest_sel.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${T}_gpu_pytest_sel.log
tail -5 gpurun_out/${T}_gpu_pytest_sel.log
python bench.py --workload cfg5mix --tile-lines 200000 --reps 5 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${T}_bench_cfg5mix_1M.json
python -c "
import json; d=json.loads(open('gpurun_out/${T}_bench_cfg5mix_1M.json').read()); g=d['gather']; print('cfg5mix 1M', round(d['value']/1e6,1), 'M lines/s gather_ms', round(d['gather_ms'],2), 'd2h', round(g['d2h_ms'],2), 'merge', round(g['merge_ms'],3), 'entries', g['entries'], 'before', g.get('entries_before_compaction'), 'table MB', round(g['table_bytes']/1e6,1))"
# the encoder's write kernel under the counters
ROOT=$(pwd); OUT=/tmp/prof_enc; rm -rf $OUT; mkdir -p $OUT
BENCH="python $ROOT/bench.py --workload cfg1 --reps 4 --steps 5 --warmup 1 --no-cpu-baseline --no-e2e --no-calib"
(cd /tmp && export TMPDIR=/tmp
 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY -d $OUT/pmc1 -o pmc1 -- $BENCH > $OUT/pmc1.log 2>&1
 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_INSTS_SMEM -d $OUT/pmc2 -o pmc2 -- $BENCH > $OUT/pmc2.log 2>&1
 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_IFETCH -d $OUT/pmc3 -o pmc3 -- $BENCH > $OUT/pmc3.log 2>&1)
for k in 'k_encode<0u, true' 'k_encode<0u, false'; do python tools/prof_summary.py $OUT "$k" > "gpurun_out/${T}_pmc_encode_$(echo $k | tr -dc 'a-z').json"; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r05f_pmc_encode_*.json')):
    d = json.load(open(f)); pm = d.get('pmc_per_dispatch_mean', {})
    print(f, {k: round(v) for k, v in pm.items()}, d.get('dispatch_info'))
PY
tail -3 $OUT/pmc2.log | cut -c1-200; tail -3 $OUT/pmc3.log | cut -c1-200
# one chunk per wave below the ticket threshold?
FG_PROBE_SIZES=1048576,4194304,16777216 FG_PROBE_OPTS=';static_chunks=1;static_chunks=1,chunk_lines=65536' python tools/probe/small_batch.py cfg2 > gpurun_out/${T}_small_cfg2_chunks.log 2>&1
FG_PROBE_SIZES=262144,524288 FG_PROBE_TOP=524288 FG_PROBE_OPTS=';static_chunks=1;static_chunks=1,chunk_lines=65536' python tools/probe/small_batch.py cfg4 cfg3 >> gpurun_out/${T}_small_cfg2_chunks.log 2>&1
grep -h "n=" gpurun_out/${T}_small_cfg2_chunks.log
for w in cfg4 cfg3 ltsv; do python tools/sweep.py $w --lines 250000 --reps 16 ';static_chunks=1;static_chunks=1,chunk_lines=65536' 2>&1 | grep "M lines/s"; done | tee gpurun_out/${T}_sweep_4M_chunks.log
