#!/bin/bash
T=${1:-r03c}
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/${T}_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${T}_pytest.log
tail -5 gpurun_out/${T}_pytest.log
python bench.py --steps 3 --warmup 1 --reps 8 2> gpurun_out/${T}_bench.err | tail -1 > gpurun_out/${T}_bench.json
tail -3 gpurun_out/${T}_bench.err
python - <<PY
import json
d=json.loads(open('gpurun_out/${T}_bench.json').read().strip().splitlines()[-1])
e=d.get('e2e',{})
print('link', e.get('link_peak',{}).get('h2d_GBps'), e.get('link_peak',{}).get('d2h_GBps'), e.get('link_peak',{}).get('bidir_GBps'), e.get('error'))
for k in ('decode_batch','frame_decode_batch','transcode_batch'):
    if k in e: print('   ', k, round(e[k]['lines_per_s']/1e6,1), 'M/s', round(e[k]['GBps_in'],1), 'GB/s in', e[k].get('frac_of_link_h2d'), e[k].get('frac_of_link_d2h'))
c=d.get('cpu_baseline',{})
print('cpu', c.get('value'), c.get('cores'), c.get('single_thread'), c.get('parallel_efficiency'), c.get('cgroup_cpu_quota'))
PY
python tools/host_path_bench.py --workload latency 2>&1 | grep batch | cut -c1-120
