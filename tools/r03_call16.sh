#!/bin/bash
# four-row form: decoder bias / logvar and ReLU-mask h requested ahead -- suite, same-box A/B, exit timeline
set -o pipefail
mkdir -p gpurun_out
ls -la 2022_cambroise_interpret_multivae_amd/*.so || exit 1
python -m pytest tests -m gpu -x -q > gpurun_out/t_r03h.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/t_r03h.log
[ $rc -eq 0 ] || exit $rc
for r in 1 2 3; do VARIANT_CONFIGS=C1 bash tools/r03_variants.sh 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/variants_pf.txt; done
python tools/exit_timeline.py C1 2>&1 | grep -v amdgpu.ids | tee gpurun_out/exit_timeline_C1_b.txt
