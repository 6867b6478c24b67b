#!/bin/bash
# The encoders of this tree against libfg_hip_r05z.so on one box (cfg1: decode -> GELF encode -> line merger, 4 M lines), then the GPU suite.
# usage (through gpurun): bash tools/r05_encode_ab.sh <tag>
tag=${1:-r05w}
out=gpurun_out
mkdir -p $out
export FG_BENCH_CACHE=/tmp/fgcache
log=$out/${tag}_ab_encode_cfg1.log
: > $log
for round in 1 2; do
  for lib in "" libfg_hip_r05z.so; do
    FLOWGGER_AMD_LIB=$lib python bench.py --workload cfg1 --steps 10 --warmup 3 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 > $out/${tag}_cfg1_${round}_${lib:-product}.json
    python - $out/${tag}_cfg1_${round}_${lib:-product}.json "${lib:-product}" >> $log <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
e = d["encode"]
print(f"cfg1 [{sys.argv[2]}] encode ms {e['ms']:.3f} lines/s {e['lines_per_s'] / 1e6:.1f} M; decode ms {d['roofline']['kernel_ms']:.3f}")
PY
  done
done
cat $log
python -m pytest tests -x -q -m gpu > $out/${tag}_gpu_pytest.log 2>&1
echo "pytest rc=$?" >> $out/${tag}_gpu_pytest.log
tail -3 $out/${tag}_gpu_pytest.log
