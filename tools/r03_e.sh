#!/bin/bash
T=${1:-r03e}
mkdir -p gpurun_out
python tools/_tc_probe.py 2>&1 | tail -9
python -m pytest tests/test_gpu_round3.py tests/test_gpu_round2.py tests/test_gpu_parity.py -m gpu -x -q -k "raw_stream or transcode or host or framed or frame or cpp or vectors or edge or pipeline or eight_threads" > gpurun_out/${T}_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${T}_pytest.log
tail -4 gpurun_out/${T}_pytest.log
python bench.py --steps 3 --warmup 1 --reps 8 --no-cpu-baseline 2> gpurun_out/${T}_bench.err | tail -1 > gpurun_out/${T}_bench.json
python - <<PY
import json
d=json.loads(open('gpurun_out/${T}_bench.json').read().strip().splitlines()[-1])
e=d.get('e2e',{})
print('link', e.get('link_peak',{}).get('h2d_GBps'), e.get('link_peak',{}).get('d2h_GBps'), e.get('link_peak',{}).get('bidir_GBps'), e.get('error'))
for k in ('decode_batch','frame_decode_batch','transcode_batch'):
    if k in e: print('   ', k, round(e[k]['lines_per_s']/1e6,1), 'M/s', round(e[k]['GBps_in'],1), 'GB/s in', e[k].get('frac_of_link_h2d'), e[k].get('frac_of_link_d2h'))
PY
