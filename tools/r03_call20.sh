#!/bin/bash
# four-row groups up to 1,024 rows (the encoder layer as a launch of its own): suite + range A/B
set -o pipefail
mkdir -p gpurun_out
ls -la 2022_cambroise_interpret_multivae_amd/*.so || exit 1
python tools/quad_range_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/quad_range_split.txt
python -m pytest tests -m gpu -x -q > gpurun_out/t_r03k.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/t_r03k.log
