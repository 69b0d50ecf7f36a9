set -o pipefail
python -m pytest tests/test_hip_large_batch.py -q > gpurun_out/t_r03i.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/t_r03i.log
python - <<'PY'
import sys, json, torch
sys.path.insert(0, ".")
import bench
print(json.dumps(bench.regime_point(torch.device("cuda", 0))), flush=True)
PY
