for n in 2048 3072; do for mr in 4096 2048; do
MOPOE_WB_MIN_ROWS=$mr python - <<PY
import os, sys, torch
sys.path.insert(0, ".")
import bench
r = bench.regime_point(torch.device("cuda", 0), n=$n)
print("rows", $n, "WB_MIN_ROWS", os.environ["MOPOE_WB_MIN_ROWS"], {k: v["avg_us"] for k, v in r["kernels"].items()}, "step %.1f us" % (1e3 * r["ms_per_step"]), flush=True)
PY
done; done 2>&1 | grep -v amdgpu.ids
