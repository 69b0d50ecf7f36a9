#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_hip_daa.py tests/test_hip_stats.py tests/test_hip_large_batch.py tests/test_hip_checkpoint.py -x -q 2>&1 | tail -4
python tools/daa_bench.py 2>&1 | grep -v amdgpu.ids | tail -6
bash tools/r03_loopfig.sh 2>&1 | grep -v amdgpu.ids
