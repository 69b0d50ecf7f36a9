set -o pipefail
python -m pytest tests/test_hip_dataset.py tests/test_hip_dp_onecall.py -q -x > gpurun_out/t_r03c.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/t_r03c.log
python - > gpurun_out/loop_r03c.txt 2>&1 <<'PY'
import sys, json, torch
sys.path.insert(0, ".")
import bench
dev = torch.device("cuda", 0)
for n in (3000, 16384):
    print(json.dumps(bench.loop_figure(dev, n)), flush=True)
PY
cat gpurun_out/loop_r03c.txt
python tools/cohort_bench.py > gpurun_out/cohort_r03c.txt 2>&1; tail -3 gpurun_out/cohort_r03c.txt
python tools/epoch_bench.py > gpurun_out/epoch_r03c.txt 2>&1; tail -3 gpurun_out/epoch_r03c.txt
python tools/host_overhead.py > gpurun_out/host_r03c.txt 2>&1; tail -15 gpurun_out/host_r03c.txt
