set -o pipefail
python -m pytest tests/test_hip_fused.py -q -x -k "specialised or C5" > gpurun_out/t_r03g.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/t_r03g.log
python - <<'PY'
import sys, json, torch
sys.path.insert(0, ".")
import bench
dev = torch.device("cuda", 0)
for k in ("C5", "C3", "C1"):
    r = bench.other_config(k, dev)
    print(k, r["ms_per_step"], r["roofline"]["kernels_avg_us"], flush=True)
PY
python tools/stage_stamps.py C5 2>&1 | tail -14
