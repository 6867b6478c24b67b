#!/bin/bash
# round 4, GPU call 3: the pair-parallel structured-data walk -- parity first, then the same-box sweep against the walker it replaces
T=${1:-r04c}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_round4.py -m gpu -x -q > gpurun_out/${T}_pytest_round4.log 2>&1; echo "round4 rc=$?"; tail -3 gpurun_out/${T}_pytest_round4.log
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py -m gpu -x -q -k "sd or cfg4 or cfg5 or long or variant or mix or overflow or fuzz or edge or error or scale" > gpurun_out/${T}_pytest_sd.log 2>&1; echo "sd rc=$?"; tail -3 gpurun_out/${T}_pytest_sd.log
timeout 300 python tools/sweep.py cfg4 --lines 250000 --reps 16 ";sd_walk=1;tile_cap=14336;tile_cap=19456;tile_cap=12288;tile_cap=24576;waves_per_cu=5;waves_per_cu=4" 2>/dev/null | grep "M lines/s" | tee gpurun_out/${T}_sweep_cfg4.log
timeout 300 python tools/sweep.py cfg5 --lines 100000 --reps 16 ";sd_walk=1;tile_cap=19456" 2>/dev/null | grep "M lines/s" | tee gpurun_out/${T}_sweep_cfg5.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY -d /tmp/pmc_sd -o pmc -- \
  python $GRAFT_REPO_ROOT/bench.py --workload cfg4 --tile-lines 250000 --reps 16 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-mix --no-calib > /tmp/pmc_sd.log 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py /tmp/pmc_sd k_rfc5424 > $GRAFT_REPO_ROOT/gpurun_out/${T}_pmc_cfg4.json 2>/dev/null
python -c "
import json; d=json.load(open('$GRAFT_REPO_ROOT/gpurun_out/${T}_pmc_cfg4.json')); p=d.get('pmc_per_dispatch_mean',{}); print({k: round(v/4e6,1) for k,v in p.items()}, 'per line')" 2>&1 | tail -1
