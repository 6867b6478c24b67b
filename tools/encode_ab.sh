#!/bin/bash
# The encoders of several builds of this tree on one box (cfg1: decode -> GELF encode -> line merger), alternated; then the encoder
# tests of the GPU suite and the counters of the product's encode kernels.
# usage (through gpurun): bash tools/encode_ab.sh <tag> "<libs: product libfg_hip_x.so ...>" [bench args, default 100 M lines]
tag=${1:-r05w}
libs=${2:-product libfg_hip_r05z.so}
shift 2
out=gpurun_out
mkdir -p $out
export FG_BENCH_CACHE=/tmp/fgcache
log=$out/${tag}_ab_encode_cfg1.log
: > $log
for round in 1 2; do
  for lib in $libs; do
    if [ "$lib" = product ]; then l=""; else l=$lib; fi
    FLOWGGER_AMD_LIB=$l python bench.py --workload cfg1 --steps 10 --warmup 3 --no-cpu-baseline --no-e2e "$@" 2>/dev/null | tail -1 > $out/${tag}_cfg1_${round}_${lib}.json
    python - $out/${tag}_cfg1_${round}_${lib}.json "$lib" >> $log <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
e = d["encode"]
n = e["lines_per_s"] * e["ms"] * 1e-3
print(f"cfg1 [{sys.argv[2]}] {n / 1e6:.0f} M lines: encode ms {e['ms']:.3f} = {e['ms'] * 4e6 / n:.3f} ms per 4 M lines, {e['lines_per_s'] / 1e6:.1f} M lines/s; decode ms {d['roofline']['kernel_ms']:.3f}")
PY
  done
done
cat $log
python -m pytest tests -x -q -m gpu -k "encod or transcode or pipeline or merger" > $out/${tag}_gpu_pytest_enc.log 2>&1
echo "pytest rc=$?" >> $out/${tag}_gpu_pytest_enc.log
tail -3 $out/${tag}_gpu_pytest_enc.log
PAT=k_encode bash tools/prof.sh ${tag}_encode --workload cfg1 --tile-lines 1000000 --reps 4 > $out/${tag}_prof_encode.log 2>&1
tail -5 $out/${tag}_prof_encode.log
