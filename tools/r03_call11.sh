python -m pytest tests/test_hip_large_batch.py -q > gpurun_out/t_r03i.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/t_r03i.log
python - <<'PY'
import os, sys, json, torch
sys.path.insert(0, ".")
import bench
r = bench.regime_point(torch.device("cuda", 0))
print({k: (v["avg_us"], v["frac_f32_mfma_peak"]) for k, v in r["kernels"].items()}, r["ms_per_step"], flush=True)
PY
