#!/bin/bash
# round 5, call 5: the whole GPU suite; the packing sink with ONE 16-byte store per span-copy step against the dword sink of rounds 1-4
# (a variant build of the same tree, alternated on this box, + a kernel trace); geometry sweep of the headline kernel; small batches
# under the final dispatch / reservation policies; the default bench line
T=${1:-r05e}
mkdir -p gpurun_out
export FG_BENCH_CACHE=/tmp/fg_bench_cache
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_gpu_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${T}_gpu_pytest.log
tail -5 gpurun_out/${T}_gpu_pytest.log
for r in 1 2; do for lib in "" libfg_hip_oldsink.so; do
  FLOWGGER_AMD_LIB=$lib python bench.py --workload cfg1 --reps 4 --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-calib 2>/dev/null | tail -1 > gpurun_out/${T}_cfg1_tmp.json
  python -c "
import json; d=json.loads(open('gpurun_out/${T}_cfg1_tmp.json').read()); e=d['encode']; print('cfg1 [${lib:-product}] encode ms', round(e['ms'],3), 'lines/s', round(e['lines_per_s']/1e6,1), 'M; decode ms', round(d['roofline']['kernel_ms'],3))"
  cp gpurun_out/${T}_cfg1_tmp.json gpurun_out/${T}_bench_cfg1_${lib:-product}.json
done; done 2>&1 | tee gpurun_out/${T}_ab_sink_cfg1.log
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/${T}_cfg1_trace -o cfg1 -- python /root/repo/bench.py --workload cfg1 --reps 4 --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-calib > /root/repo/gpurun_out/${T}_cfg1_trace.log 2>&1)
python - <<'PY' | tee gpurun_out/r05e_cfg1_kernels.log
import sqlite3, glob, collections
for f in glob.glob('gpurun_out/r05e_cfg1_trace/*.db'):
    con = sqlite3.connect(f)
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]; ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
    by = collections.defaultdict(list)
    for name, s, e in con.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id where s.kernel_name like '%fg%'"):
        by[name[:70]].append((e - s) / 1e3)
    for k, v in sorted(by.items()):
        v = sorted(v); print(k, len(v), 'launches, median us', round(v[len(v) // 2], 1))
PY
python tools/sweep.py cfg2 --lines 1000000 --reps 40 ';lines_per_group=48;lines_per_group=56;lines_per_group=32;waves_per_cu=6;tile_cap=18432;tile_cap=20480;chunk_lines=512;' 2>&1 | grep "M lines/s" | tee gpurun_out/${T}_sweep_cfg2_40M.log
export FG_PROBE_SIZES=65536,262144,1048576
python tools/probe/small_batch.py cfg3 ltsv > gpurun_out/${T}_small.log 2>&1
FG_PROBE_SIZES=1048576,4194304,16777216 FG_PROBE_OPTS=';static_chunks=1' python tools/probe/small_batch.py cfg2 >> gpurun_out/${T}_small.log 2>&1
grep -h "n=" gpurun_out/${T}_small.log
env -u FG_BENCH_CACHE python bench.py 2> gpurun_out/${T}_bench_default.err | tail -1 > gpurun_out/${T}_bench_default_100M.json
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r05e_bench_default_100M.json').read())
r = d['roofline']
print('headline', round(d['value'] / 1e9, 2), 'G lines/s frac', round(r['frac'], 3), 'of copy', round(r.get('frac_of_copy', 0), 3))
for k in ('configs2', 'configs3', 'configs4'):
    c = d.get(k, {})
    print(k, round(c.get('value', 0) / 1e9, 3), 'G', c.get('roofline_frac'), c.get('gather_ms'), {kk: round(v['lines_per_s'] / 1e6, 1) for kk, v in (c.get('e2e') or {}).items() if isinstance(v, dict) and 'lines_per_s' in v})
print('small_batch', {k: {n: round(v['lines_per_s'] / 1e6, 1) for n, v in vv.items()} for k, vv in d.get('small_batch', {}).items() if isinstance(vv, dict)})
print('e2e', {k: round(v['lines_per_s'] / 1e6, 1) for k, v in (d.get('e2e') or {}).items() if isinstance(v, dict) and 'lines_per_s' in v})
PY
