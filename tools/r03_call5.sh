set -o pipefail
python -m pytest tests/test_hip_fused.py tests/test_hip_parity.py tests/test_hip_trajectory.py tests/test_hip_stats.py -q -x > gpurun_out/t_r03f.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/t_r03f.log
python - <<'PY'
import sys, json, torch
sys.path.insert(0, ".")
import bench
dev = torch.device("cuda", 0)
for k in ("C5", "C3", "C1"):
    r = bench.other_config(k, dev)
    print(k, r["ms_per_step"], r["roofline"]["kernels_avg_us"], flush=True)
PY
