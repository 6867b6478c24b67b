#!/bin/bash
# round 5, call 7: the encoders with their rare global-memory path out of line (no stack copies of the kernel arguments in the hot path)
T=${1:-r05g}
mkdir -p gpurun_out
export FG_BENCH_CACHE=/tmp/fg_bench_cache
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -x -q -k "encod or transcode or pipeline" > gpurun_out/${T}_gpu_pytest_enc.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${T}_gpu_pytest_enc.log
tail -4 gpurun_out/${T}_gpu_pytest_enc.log
for r in 1 2; do for lib in "" libfg_hip_oldsink.so; do
  FLOWGGER_AMD_LIB=$lib python bench.py --workload cfg1 --reps 4 --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-calib 2>/dev/null | tail -1 > gpurun_out/${T}_cfg1_tmp.json
  python -c "
import json; d=json.loads(open('gpurun_out/${T}_cfg1_tmp.json').read()); e=d['encode']; print('cfg1 [${lib:-product}] encode ms', round(e['ms'],3), 'lines/s', round(e['lines_per_s']/1e6,1), 'M; decode ms', round(d['roofline']['kernel_ms'],3))"
  cp gpurun_out/${T}_cfg1_tmp.json gpurun_out/${T}_bench_cfg1_${lib:-product}.json
done; done 2>&1 | tee gpurun_out/${T}_ab_encode_cfg1.log
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/${T}_cfg1_trace -o cfg1 -- python /root/repo/bench.py --workload cfg1 --reps 4 --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-calib > /root/repo/gpurun_out/${T}_cfg1_trace.log 2>&1)
python - <<'PY' | tee gpurun_out/r05g_cfg1_kernels.log
import sqlite3, glob, collections
for f in glob.glob('gpurun_out/r05g_cfg1_trace/*.db'):
    con = sqlite3.connect(f)
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]; ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
    by = collections.defaultdict(list)
    for name, s, e in con.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id where s.kernel_name like '%fg%'"):
        by[name[:70]].append((e - s) / 1e3)
    for k, v in sorted(by.items()):
        v = sorted(v); print(k, len(v), 'launches, median us', round(v[len(v) // 2], 1))
PY
# the other encoders / sources (kernel times through fg_last_kernel_ms would need code: the cfg1 pipeline on the SD corpus instead)
FG_PROBE_SIZES=1048576,4194304,8388608,16777216 FG_PROBE_OPTS=';static_chunks=1' python tools/probe/small_batch.py cfg2 > gpurun_out/${T}_small_cfg2.log 2>&1
grep -h "n=" gpurun_out/${T}_small_cfg2.log
