#!/bin/bash
# first GPU call of round 4 (about a minute): where the time of the sliced host path goes for the corpora with entries -- per-slice
# upload / kernel / download spans of the library's own pipeline shape, replayed outside the library with timing events
# (tools/probe/e2e_timeline.cpp).  Output: gpurun_out/<tag>_timeline_<corpus>_<mode>.log
T=${1:-r04a}
mkdir -p gpurun_out
python - <<'PY'
from flowgger_amd import synth
n = 2_000_000
open("/tmp/tl_gelf.txt", "wb").write(b"\n".join(ln for ln in synth.gelf_lines(n) if b"\n" not in ln))
open("/tmp/tl_ltsv.txt", "wb").write(b"\n".join(ln for ln in synth.ltsv_lines(n) if b"\n" not in ln))
open("/tmp/tl_cfg4.txt", "wb").write(b"\n".join(synth.rfc5424_lines(n, cfg=4, sd=True)))
open("/tmp/tl_cfg2.txt", "wb").write(b"\n".join(synth.rfc5424_lines(n, cfg=2)))
PY
for c in "gelf gelf" "ltsv ltsv" "rfc5424 cfg4" "rfc5424 cfg2"; do set -- $c
  for m in all collect; do ./tools/probe/e2e_timeline $1 /tmp/tl_$2.txt 32 $m > gpurun_out/${T}_timeline_$2_$m.log 2>&1; head -1 gpurun_out/${T}_timeline_$2_$m.log; tail -1 gpurun_out/${T}_timeline_$2_$m.log; done
done
