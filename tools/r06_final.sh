#!/bin/bash
# The round's closing measurements on ONE MI355X box (sources of the final commit):
#   GPU suite + smoke; the default bench line as the driver runs it; a kernel trace (--stats, no counters) of the SAME 100 M-line
#   workload beside its JSON line (so that profile-derived and line-derived fractions reconcile: VERDICT r4 item 6a, kept in round 6); kernel trace +
#   PMC + HBM traffic of the headline at 100 M lines on this box (6b); HBM traffic of every other workload (FETCH_SIZE / WRITE_SIZE in
#   separate passes); instruction counters of the structured-data and GELF kernels; the BASELINE configurations at full size; the
#   PCIe-inclusive legs of the entry corpora; the latency table (6c).
# FG_BENCH_CACHE: the generated tiles are pickled once (the generators are deterministic) -- only the measurement scripts use it.
T=${1:-r06fin}
mkdir -p gpurun_out
export FG_BENCH_CACHE=/tmp/fg_bench_cache
python -m pytest tests -m gpu -x -q > gpurun_out/${T}_gpu_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${T}_gpu_pytest.log
tail -3 gpurun_out/${T}_gpu_pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
env -u FG_BENCH_CACHE python bench.py 2> gpurun_out/${T}_bench_default.err | tail -1 > gpurun_out/${T}_bench_default_100M.json
cut -c1-200 gpurun_out/${T}_bench_default_100M.json
# the same 100 M-line main workload under rocprofv3 --kernel-trace --stats (no counters): kernel average vs roofline.kernel_ms of ITS line
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt_$T && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$T -o kt -- python /root/repo/bench.py --gpus 1 --steps 20 --warmup 5 --no-legs --no-mix --no-e2e --no-cpu-baseline > /tmp/kt_$T.log 2>&1)
grep -h '"metric"' /tmp/kt_$T.log | tail -1 > gpurun_out/${T}_driver_cmd_bench_line.json
python - <<PY
import csv, glob, json
rows = []
for f in glob.glob('/tmp/kt_$T/**/*kernel_stats.csv', recursive=True):
    rows += [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: -float(r.get('TotalDurationNs', 0)))
line = json.loads(open('gpurun_out/${T}_driver_cmd_bench_line.json').read())
k = [r for r in rows if 'k_rfc5424' in r['Name']][0]
out = {'command': 'rocprofv3 --kernel-trace --stats -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-legs --no-mix --no-e2e --no-cpu-baseline',
       'kernel_stats': [{kk: r[kk] for kk in ('Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs') if kk in r} for r in rows[:8]],
       'headline_kernel_avg_ms_rocprof': float(k['AverageNs']) / 1e6, 'bench_line_kernel_ms': line['roofline']['kernel_ms'],
       'bench_line_frac': line['roofline']['frac'],
       'frac_from_rocprof': line['roofline']['algorithmic_bytes_per_launch'] / (float(k['AverageNs']) * 1e-9) / 1e9 / 8000.0}
json.dump(out, open('gpurun_out/${T}_driver_cmd_kernel_trace.json', 'w'), indent=1)
print('rocprof avg ms', round(out['headline_kernel_avg_ms_rocprof'], 4), 'line kernel_ms', round(out['bench_line_kernel_ms'], 4), 'frac', round(out['frac_from_rocprof'], 4), 'vs', round(out['bench_line_frac'], 4))
PY
# headline: kernel trace + PMC + traffic at 100 M lines on this box
PAT=k_rfc5424 bash tools/prof.sh ${T}_cfg2 --reps 100 --no-mix --no-legs --no-calib > gpurun_out/${T}_prof_cfg2.log 2>&1
# HBM traffic of the other workloads (4 M lines each)
# (cfg3 / cfg4: the FULL set of passes -- kernel trace, instruction counters, traffic -- so that tools/update_traffic.py can stamp the compute
#  roof beside the traffic: VERDICT r5 item 5)
PAT='k_gelf<' bash tools/prof.sh ${T}_cfg3 --workload cfg3 --tile-lines 250000 --reps 16 --no-calib > gpurun_out/${T}_prof_cfg3.log 2>&1
PAT=k_rfc5424 bash tools/prof.sh ${T}_cfg4 --workload cfg4 --tile-lines 250000 --reps 16 --no-calib > gpurun_out/${T}_prof_cfg4.log 2>&1
bash tools/prof_traffic.sh ${T}_cfg5 k_rfc5424 --workload cfg5 --tile-lines 250000 --reps 16 --no-calib > /dev/null 2>&1
bash tools/prof_traffic.sh ${T}_ltsv k_ltsv --workload ltsv --tile-lines 250000 --reps 16 --no-calib > /dev/null 2>&1
bash tools/prof_traffic.sh ${T}_ltsv5 k_ltsv --workload ltsv5 --tile-lines 250000 --reps 16 --no-calib > /dev/null 2>&1
# instruction counters of the long-tail kernel
PAT=k_rfc5424 bash tools/prof_quick.sh ${T}_cfg5 k_rfc5424 --workload cfg5 --tile-lines 250000 --reps 16 --no-calib > /dev/null 2>&1
# the BASELINE configurations at full size (250 K-line tiles replicated)
python bench.py --workload cfg3 --tile-lines 250000 --reps 400 --steps 10 --warmup 2 --no-e2e 2>/dev/null | tail -1 > gpurun_out/${T}_bench_cfg3_100M.json
python bench.py --workload cfg4 --tile-lines 250000 --reps 500 --steps 5 --warmup 1 --no-e2e 2>/dev/null | tail -1 > gpurun_out/${T}_bench_cfg4_125M.json
python bench.py --workload cfg5 --tile-lines 250000 --reps 160 --steps 5 --warmup 1 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 > gpurun_out/${T}_bench_cfg5_40M.json
python bench.py --workload ltsv --tile-lines 250000 --reps 400 --steps 10 --warmup 2 --no-e2e 2>/dev/null | tail -1 > gpurun_out/${T}_bench_ltsv_100M.json
python bench.py --workload ltsv5 --tile-lines 250000 --reps 80 --steps 5 --warmup 1 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 > gpurun_out/${T}_bench_ltsv5_20M.json
python bench.py --workload cfg5mix 2>/dev/null | tail -1 > gpurun_out/${T}_bench_cfg5mix.json
python bench.py --workload rfc3164 --tile-lines 250000 --reps 400 --steps 10 --warmup 2 --no-e2e --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${T}_bench_rfc3164_100M.json
python bench.py --workload frame --steps 5 --warmup 1 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 > gpurun_out/${T}_bench_frame.json
python bench.py --workload cfg1 --reps 4 --steps 5 --warmup 2 2>/dev/null | tail -1 > gpurun_out/${T}_bench_cfg1_pipeline.json
# the PCIe-inclusive legs of the entry corpora
for w in cfg3 ltsv cfg4; do python bench.py --workload $w --tile-lines 250000 --reps 16 --steps 5 --warmup 2 --no-cpu-baseline --no-calib 2>/dev/null | tail -1 > gpurun_out/${T}_e2e_$w.json; done
for f in bench_default_100M bench_cfg3_100M bench_cfg4_125M bench_cfg5_40M bench_ltsv_100M bench_ltsv5_20M bench_cfg5mix bench_rfc3164_100M bench_frame bench_cfg1_pipeline e2e_cfg3 e2e_ltsv e2e_cfg4; do python -c "
import json; d=json.loads(open('gpurun_out/${T}_$f.json').read().strip().splitlines()[-1]); r=d['roofline']; e=d.get('e2e') or {}
print('$f', round(d['value']/1e6,1), 'M lines/s', round(r['kernel_ms'],3), 'ms frac', round(r['frac'],4), 'read', round(r.get('read_only_frac',0),4), 'of copy', r.get('frac_of_copy'), 'cpu', d.get('cpu_baseline',{}).get('value'), d.get('gather_ms'), d.get('framing',{}).get('GBps'), d.get('encode',{}).get('ms'), {k: round(v/1e6,1) for k,v in (e.get('aggregate') or {}).items()})" 2>&1 | tail -1; done
# round 6: framing inside the decode kernels -- resident (two-step vs fused, one box) and from pinned host memory (variants alternated)
python tools/probe/fused_frame.py cfg2 cfg4 ltsv cfg3 2>/dev/null | grep "^{" > gpurun_out/${T}_fused_frame.log
FG_PROBE_OPTS=";no_fused_framing=1" python tools/probe/fused_host.py cfg2 cfg3 cfg4 ltsv 2>/dev/null | grep "^{" > gpurun_out/${T}_fused_host.log
FG_PROBE_LINES=250000 FG_PROBE_OPTS=";no_fused_framing=1" python tools/probe/fused_host.py cfg2 cfg3 cfg4 2>/dev/null | grep "^{" > gpurun_out/${T}_fused_host_250K.log
# round 6: RFC3164, the slow shapes regrouped -- same box, alternated
for m in 0 2 0 2; do python bench.py --workload rfc3164 --tile-lines 250000 --reps 400 --steps 10 --warmup 2 --no-e2e --no-cpu-baseline --no-calib --launch-opts rfc3164_regroup=$m 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('rfc3164 100 M lines, rfc3164_regroup=$m (0 = the library: regrouped, 2 = plain kernel):', round(d['value']/1e9,3), 'G lines/s', round(d['roofline']['kernel_ms'],3), 'ms')"; done > gpurun_out/${T}_rfc3164_regroup_ab.log
python tools/probe/rfc3164_shapes.py 2>/dev/null | grep rfc3164 > gpurun_out/${T}_rfc3164_shapes.log
# the latency table: microseconds per fg_decode_batch call against the batch size (zero-copy form, pinned buffers)
python tools/host_path_bench.py --workload latency 2>/dev/null | tee gpurun_out/${T}_latency_sweep.log | grep "^batch"
ls gpurun_out | grep -c $T
