#!/bin/bash
# evidence for DESIGN 4.0's claims, one box: throughput vs waves per CU / tile / chunk (tools/sweep.py), which stream pairs overlap the two
# copy directions, rocprofv3 kernel statistics of the entry workloads
T=${1:-r03z}
mkdir -p gpurun_out
{ python tools/sweep.py cfg4 ";waves_per_cu=2;waves_per_cu=3;waves_per_cu=4;waves_per_cu=5;waves_per_cu=6;tile_cap=32768;chunk_lines=256;chunk_lines=1024" 2>/dev/null
  python tools/sweep.py ltsv ";waves_per_cu=4;waves_per_cu=5;waves_per_cu=6;tile_cap=12288;tile_cap=24576" 2>/dev/null
  python tools/sweep.py cfg5 ";no_head=1;waves_per_cu=4;tile_cap=32768" 2>/dev/null
  python tools/sweep.py cfg2 ";chunk_lines=1024;chunk_lines=65536;waves_per_cu=5" 2>/dev/null; } | grep "M lines/s" > gpurun_out/${T}_sweep.log
cat gpurun_out/${T}_sweep.log
./tools/probe/stream_pairs > gpurun_out/${T}_stream_pairs.log 2>&1; tail -8 gpurun_out/${T}_stream_pairs.log
cd /tmp && export TMPDIR=/tmp
for w in cfg4 ltsv cfg5; do
  rm -rf /tmp/ks_$w; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$w -o ks -- python $GRAFT_REPO_ROOT/bench.py --workload $w --tile-lines 1000000 --reps 4 --steps 5 --warmup 1 --no-cpu-baseline --no-e2e > /tmp/ks_$w.log 2>&1
  f=$(find /tmp/ks_$w -name "*kernel_stats.csv" | head -1); head -4 "$f" | cut -c1-200 > $GRAFT_REPO_ROOT/gpurun_out/${T}_kernel_stats_$w.csv; grep -h '"metric"' /tmp/ks_$w.log | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$w live kernel_ms', d['roofline']['kernel_ms'])" >> $GRAFT_REPO_ROOT/gpurun_out/${T}_kernel_stats_$w.csv
  cat $GRAFT_REPO_ROOT/gpurun_out/${T}_kernel_stats_$w.csv
done
