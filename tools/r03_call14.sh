#!/bin/bash
# pass B again (the library was missing from the snapshot) + store-acknowledgement knock-outs
set -o pipefail
mkdir -p gpurun_out
ls -la 2022_cambroise_interpret_multivae_amd/*.so || exit 1
python -m pytest tests -m gpu -x -q > gpurun_out/t_r03f.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/t_r03f.log
[ $rc -eq 0 ] || exit $rc
python - <<'PY'
import sys, torch
sys.path.insert(0, ".")
import bench
r = bench.regime_point(torch.device("cuda", 0))
print({k: (v["avg_us"], v["frac_f32_mfma_peak"]) for k, v in r["kernels"].items()}, r["ms_per_step"], flush=True)
PY
KNOCK_STEPS=1000 KNOCK_MASKS="no global result stores=0x400000;no stores, no S3 weights on the spot (S3 out)=0x400040;no stores, exit after S2b=0x5400000;exit after S2b=0x5000000;no stores, exit after S3=0x7400000;exit after S3=0x7000000;no stores, exit after S5=0xa400000;exit after S5=0xa000000" python tools/knockout.py 0 C1 > gpurun_out/knock_stores_C1.txt 2>&1; cat gpurun_out/knock_stores_C1.txt
ROUNDS=2 STEPS=2000 bash tools/ab_env.sh HIP_FORCE_DEV_KERNARG=0 HIP_FORCE_DEV_KERNARG=1 > gpurun_out/ab_kernarg.txt 2>&1; cat gpurun_out/ab_kernarg.txt
TAG=r03e bash tools/r03_profiles.sh C1 C3 C5
timeout -k 10 500 bash tools/profile_round.sh r03e_N64K --config N64K --steps 30 --warmup 5 --settle 0 --no-cpu-baseline --no-roofline --quick > gpurun_out/prof_r03e_N64K.log 2>&1; echo "profile N64K rc=$?"; tail -12 gpurun_out/prof_r03e_N64K.log
