#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for r in 1 2; do VARIANT_CONFIGS=C5,C3,C1 bash tools/r03_variants.sh 2>&1 | grep -v amdgpu.ids; done
for lib in libmopoe_hip.so libmopoe_hip_vLDSB2.so libmopoe_hip.so libmopoe_hip_vLDSB2.so; do
MOPOE_LIB=$lib python - <<'PY'
import os, sys, torch
sys.path.insert(0, ".")
import bench
r = bench.regime_point(torch.device("cuda", 0))
print(os.environ["MOPOE_LIB"], {k: (v["avg_us"], v["frac_f32_mfma_peak"]) for k, v in r["kernels"].items()}, r["ms_per_step"], flush=True)
PY
done 2>&1 | grep -v amdgpu.ids
MOPOE_LIB=libmopoe_hip_vLDSB2.so python -m pytest tests -m gpu -x -q > gpurun_out/t_r03o.log 2>&1; echo "pytest (LDSB2) rc=$?"; tail -3 gpurun_out/t_r03o.log
