#!/bin/bash
# round 5, probe 1: kernel-only time against the batch size (the small-batch collapse of VERDICT r4 item 2) + what the long-tail RFC5424
# kernel's fixed cost is made of (lines whose structured data runs past the 1 KiB head are parsed from global memory, a dword per trip)
T=${1:-r05a}
mkdir -p gpurun_out
export FG_BENCH_CACHE=/tmp/fg_bench_cache
export FG_PROBE_SIZES=16384,65536,262144,524288,1048576
python tools/probe/small_batch.py cfg2 cfg5 ltsv5 cfg4 cfg3 ltsv > gpurun_out/${T}_small.log 2>&1
FG_PROBE_SIZES=65536,524288 FG_PROBE_FILTER=sd900 python tools/probe/small_batch.py cfg5 > gpurun_out/${T}_small_cfg5_sd900.log 2>&1
FG_PROBE_SIZES=65536,524288 FG_PROBE_INVALID=0 python tools/probe/small_batch.py cfg5 > gpurun_out/${T}_small_cfg5_noinv.log 2>&1
FG_PROBE_SIZES=65536,524288 FG_PROBE_INVALID=0 FG_PROBE_FILTER=sd900 python tools/probe/small_batch.py cfg5 > gpurun_out/${T}_small_cfg5_noinv_sd900.log 2>&1
FG_PROBE_SIZES=524288 FG_PROBE_OPTS=';waves_per_cu=4;waves_per_cu=6;chunk_lines=64;chunk_lines=128;no_head=1' python tools/probe/small_batch.py cfg5 > gpurun_out/${T}_small_cfg5_opts.log 2>&1
FG_PROBE_SIZES=524288 FG_PROBE_OPTS=';waves_per_cu=4;chunk_lines=64;chunk_lines=128;no_head=1' python tools/probe/small_batch.py ltsv5 > gpurun_out/${T}_small_ltsv5_opts.log 2>&1
grep -h "n=" gpurun_out/${T}_small*.log | grep -v "^#"
