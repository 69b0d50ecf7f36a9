#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
ls -la 2022_cambroise_interpret_multivae_amd/*.so || exit 1
python tools/quad_range_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/quad_range_split_e.txt
for v in 2048 1000000; do
MOPOE_LIN_BIG_ROWS=$v python - <<'PY'
import os, sys, torch
sys.path.insert(0, ".")
import bench
r = bench.regime_point(torch.device("cuda", 0))
print("LIN_BIG_ROWS", os.environ["MOPOE_LIN_BIG_ROWS"], {k: (v["avg_us"], v["frac_f32_mfma_peak"]) for k, v in r["kernels"].items()}, r["ms_per_step"], flush=True)
PY
done 2>&1 | grep -v amdgpu.ids
