#!/bin/bash
# Round 4, first GPU call (about four minutes): (1) where the sliced host path's time goes for the corpora with entries
# (tools/r04_timeline.sh); (2) branch r4-sd2's kernels (flowgger_amd/libfg_hip_r4sd2.so, built from the branch beside the tree's own
# library -- rebuild it when the branch moves: checkout, python -m flowgger_amd.build, copy) against this tree's ON ONE BOX: parity of
# the structured-data tests first, then throughput under both group sizes, then VALU / LDS-wait counters of both.
T=${1:-r04a}
mkdir -p gpurun_out
bash tools/r04_timeline.sh $T 2>&1 | tail -20
B=libfg_hip_r4sd2.so
if [ -f flowgger_amd/$B ]; then
  FLOWGGER_AMD_LIB=$B python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py -m gpu -x -q -k "sd or cfg4 or cfg5 or long or variant or mix or overflow" > gpurun_out/${T}_pytest_branch.log 2>&1
  tail -2 gpurun_out/${T}_pytest_branch.log
  for lib in "" $B; do
    echo "== ${lib:-this tree}"
    FLOWGGER_AMD_LIB=$lib python tools/sweep.py cfg4 ";lines_per_group=32;lines_per_group=32,waves_per_cu=6" 2>/dev/null | grep "M lines/s"
    FLOWGGER_AMD_LIB=$lib python tools/sweep.py cfg5 ";lines_per_group=32" 2>/dev/null | grep "M lines/s"
  done | tee gpurun_out/${T}_ab_sweep.log
  cd /tmp && export TMPDIR=/tmp
  for lib in "" $B; do
    tag=${lib:-main}; tag=${tag%.so}
    FLOWGGER_AMD_LIB=$lib rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY -d /tmp/ab_$tag -o ab -- \
      python $GRAFT_REPO_ROOT/bench.py --workload cfg4 --tile-lines 1000000 --reps 4 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-mix > /tmp/ab_$tag.log 2>&1
    python $GRAFT_REPO_ROOT/tools/prof_summary.py /tmp/ab_$tag k_rfc5424 > $GRAFT_REPO_ROOT/gpurun_out/${T}_ab_pmc_$tag.json 2>/dev/null
    python -c "
import json; d=json.load(open('$GRAFT_REPO_ROOT/gpurun_out/${T}_ab_pmc_$tag.json')); p=d.get('pmc_per_dispatch_mean',{}); print('$tag', {k: round(v/4e6,1) for k,v in p.items() if k.startswith('SQ_INSTS') or 'WAIT' in k}, 'per line')" 2>&1 | tail -1
  done
fi
