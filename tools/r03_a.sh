#!/bin/bash
# round 3, first GPU call: the suite, smoke, the default bench line, cfg5mix, the self-launched ranks path
T=${1:-r03a}
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/${T}_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${T}_pytest.log
tail -5 gpurun_out/${T}_pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py 2> gpurun_out/${T}_bench_default.err | tail -1 > gpurun_out/${T}_bench_default_100M.json
tail -5 gpurun_out/${T}_bench_default.err
cut -c1-400 gpurun_out/${T}_bench_default_100M.json
python bench.py --workload cfg5mix 2> gpurun_out/${T}_bench_cfg5mix.err | tail -1 > gpurun_out/${T}_bench_cfg5mix.json
tail -5 gpurun_out/${T}_bench_cfg5mix.err
python bench.py --spawn --steps 3 --warmup 1 --reps 4 --no-cpu-baseline 2> gpurun_out/${T}_bench_spawn.err | tail -1 > gpurun_out/${T}_bench_spawn.json
tail -5 gpurun_out/${T}_bench_spawn.err
python bench.py --spawn --workload cfg5mix --steps 3 --warmup 1 --reps 2 --no-cpu-baseline 2> gpurun_out/${T}_bench_spawn5.err | tail -1 > gpurun_out/${T}_bench_spawn5.json
tail -5 gpurun_out/${T}_bench_spawn5.err
for f in default_100M cfg5mix spawn spawn5; do python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/${T}_bench_$f.json').read().strip().splitlines()[-1])
    r=d['roofline']; e=d.get('e2e',{})
    print('$f', round(d['value']/1e6,1), 'M lines/s', round(r['kernel_ms'],3), 'ms frac', round(r['frac'],4), 'cpu', d.get('cpu_baseline',{}).get('value'), d.get('cpu_baseline',{}).get('single_thread'), d.get('cpu_baseline',{}).get('parallel_efficiency'))
    print('   e2e', e.get('aggregate'), e.get('link_peak',{}).get('h2d_GBps'), e.get('link_peak',{}).get('d2h_GBps'), e.get('link_peak',{}).get('bidir_GBps'), e.get('error'))
    for k in ('decode_batch','frame_decode_batch','transcode_batch'):
        if k in e: print('   ', k, round(e[k]['lines_per_s']/1e6,1), 'M/s', round(e[k]['GBps_in'],1), 'GB/s in', e[k].get('frac_of_link_h2d'), e[k].get('frac_of_link_d2h'))
    print('   gather', d.get('gather'), d.get('sub_batches'), d['ranks'].get('numa'))
except Exception as ex:
    print('$f', 'ERR', ex)
PY
done
