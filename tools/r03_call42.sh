#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_hip_large_batch.py -x -q 2>&1 | tail -3
python - <<'PY'
import sys, torch
sys.path.insert(0, ".")
import bench
for op in ("f32", "bf16", "f32", "bf16"):
    r = bench.regime_point(torch.device("cuda", 0), operands=op)
    print(op, {k: (v["avg_us"], v["frac_f32_mfma_peak"]) for k, v in r["kernels"].items()}, r["ms_per_step"], flush=True)
PY
