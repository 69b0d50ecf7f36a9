#!/bin/bash
# big weight-gradient step: parallel slice reduce + folded slabs; finer exit timeline
set -o pipefail
mkdir -p gpurun_out
ls -la 2022_cambroise_interpret_multivae_amd/*.so || exit 1
python -m pytest tests/test_hip_large_batch.py tests/test_hip_parity.py tests/test_hip_fused.py -x -q > gpurun_out/t_r03i.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/t_r03i.log
[ $rc -eq 0 ] || exit $rc
python - <<'PY'
import sys, torch
sys.path.insert(0, ".")
import bench
r = bench.regime_point(torch.device("cuda", 0))
print({k: (v["avg_us"], v["frac_f32_mfma_peak"]) for k, v in r["kernels"].items()}, r["ms_per_step"], flush=True)
PY
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_n64k_b -o run -- python3 bench.py --config N64K --steps 30 --warmup 5 --settle 0 --no-cpu-baseline --no-roofline --quick > gpurun_out/n64k_b.log 2>&1; echo "rocprof rc=$?"
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_n64k_b/**/*kernel_stats.csv", recursive=True)
for row in list(csv.DictReader(open(f[0])))[:9]:
    print("%-90s %5s %10.1f us" % (row["Name"][:90], row["Calls"], float(row["AverageNs"]) / 1e3))
PY
python tools/exit_timeline.py C1 2>&1 | grep -v amdgpu.ids | tee gpurun_out/exit_timeline_C1_c.txt
