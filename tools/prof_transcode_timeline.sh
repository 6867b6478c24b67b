#!/bin/bash
# usage: [OPTS=k=v,k=v] tools/prof_transcode_timeline.sh <tag> [tiles = 4] -- kernel + copy traces of fg_transcode_batch (no counters), condensed into
# gpurun_out/<tag>_transcode_timeline.json by tools/probe/transcode_timeline.py
TAG=$1; TILES=${2:-4}
R=$(pwd)
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tt_$TAG
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tt_$TAG -o tt -- python $R/tools/probe/transcode_timeline.py run $TILES "$OPTS" > /tmp/tt_$TAG.log 2>&1
cd $R
python tools/probe/transcode_timeline.py run $TILES "$OPTS" 2>/dev/null | grep "^{" | sed 's/^/untraced: /' >> /tmp/tt_$TAG.log
python tools/probe/transcode_timeline.py report /tmp/tt_$TAG /tmp/tt_$TAG.log > gpurun_out/${TAG}_transcode_timeline.json
grep "untraced" /tmp/tt_$TAG.log
head -c ${3:-2500} gpurun_out/${TAG}_transcode_timeline.json
