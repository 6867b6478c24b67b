#!/bin/bash
# round 4, GPU call 12: the raw-stream host path uploaded by the framing scan itself (pinned chunks): parity of the frame tests, e2e legs
T=${1:-r04l}
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "fram or stream or transcode or splitter or host_path or chunk" > gpurun_out/${T}_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/${T}_pytest.log
for w in cfg2 cfg3 ltsv cfg4; do
  timeout 300 python bench.py --workload $w --tile-lines 250000 --reps 16 --steps 5 --warmup 2 --no-cpu-baseline --no-mix --no-legs --no-calib > gpurun_out/${T}_bench_$w.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/${T}_bench_$w.json").read().strip().splitlines()[-1]); e=d.get("e2e",{})
print("$w", round(d["value"]/1e6,1), "M lines/s kernel;", {k: (round(e[k]["lines_per_s"]/1e6,1), round(e[k].get("frac_of_link_h2d") or 0,3)) for k in ("decode_batch","frame_decode_batch","transcode_batch") if k in e and "lines_per_s" in e[k]})
PY
done
