#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
ls -la 2022_cambroise_interpret_multivae_amd/*.so || exit 1
for lib in libmopoe_hip.so libmopoe_hip_vBIGK16.so libmopoe_hip.so libmopoe_hip_vBIGK16.so; do
MOPOE_LIB=$lib python - <<'PY'
import os, sys, torch
sys.path.insert(0, ".")
import bench
r = bench.regime_point(torch.device("cuda", 0))
print(os.environ["MOPOE_LIB"], {k: (v["avg_us"], v["frac_f32_mfma_peak"]) for k, v in r["kernels"].items()}, r["ms_per_step"], flush=True)
PY
done 2>&1 | grep -v amdgpu.ids
python tools/quad_range_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/quad_range_split_d.txt
RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 python bench.py --force-dist --quick --no-cpu-baseline --steps 1000 --warmup 100 > gpurun_out/bench_r03h_dp1_rccl.json 2> gpurun_out/bench_r03h_dp1.err; echo "dp1 rc=$?"; wc -l gpurun_out/bench_r03h_dp1_rccl.json
