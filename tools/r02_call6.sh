#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -k "gelf or Gelf or entry_workloads or reference_vectors or transcode or encoder" > gpurun_out/r02e_pytest_gelf.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02e_pytest_gelf.log
tail -4 gpurun_out/r02e_pytest_gelf.log
LS="8 12 16 24 32" bash tools/r02_prof_gelf.sh
