#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_hip_fused.py tests/test_hip_parity.py -x -q 2>&1 | tail -3
VARIANT_CONFIGS=C5,C3,C1 bash tools/r03_variants.sh 2>&1 | grep -v amdgpu.ids
VARIANT_CONFIGS=C5,C3,C1 bash tools/r03_variants.sh 2>&1 | grep -v amdgpu.ids
