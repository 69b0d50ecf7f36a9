#!/bin/bash
# four-row form: stage barriers that wait for LDS only -- suite, same-box A/B
set -o pipefail
mkdir -p gpurun_out
ls -la 2022_cambroise_interpret_multivae_amd/*.so || exit 1
for r in 1 2 3; do VARIANT_CONFIGS=C1 bash tools/r03_variants.sh 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/variants_sync.txt; done
python -m pytest tests -m gpu -x -q > gpurun_out/t_r03j.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/t_r03j.log
