set -o pipefail
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
python -m pytest tests/test_hip_dp_onecall.py tests/test_hip_invalid.py tests/test_hip_xgmi.py tests/test_hip_topology.py -q -x > gpurun_out/t_r03e.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/t_r03e.log
python bench.py --force-dist --quick --no-cpu-baseline --steps 2000 --warmup 200 2> gpurun_out/b_r03e_dp1_rccl.err | grep '^{' > gpurun_out/b_r03e_dp1_rccl.json; echo "rccl rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/b_r03e_dp1_rccl.json")); print(d["ms_per_step"], d["roofline"]["kernels_avg_us"])
PY
