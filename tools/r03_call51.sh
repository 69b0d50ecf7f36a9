#!/bin/bash
for lib in libmopoe_hip.so libmopoe_hip_vWBc18.so libmopoe_hip_vWBc24.so libmopoe_hip.so libmopoe_hip_vWBc18.so libmopoe_hip_vWBc24.so; do
MOPOE_LIB=$lib python - <<'PY'
import os, sys, torch
sys.path.insert(0, ".")
import bench
r = bench.regime_point(torch.device("cuda", 0))
print(os.environ["MOPOE_LIB"], {k: (v["avg_us"]) for k, v in r["kernels"].items()}, r["ms_per_step"], flush=True)
PY
done 2>&1 | grep -v amdgpu.ids
python -m pytest tests/test_hip_large_batch.py -x -q 2>&1 | tail -2
