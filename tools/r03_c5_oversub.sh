for fb in 256 512 384 256 512; do
MOPOE_FUSE_BLOCKS=$fb python - <<'PY'
import os, sys, torch
sys.path.insert(0, ".")
import bench
r = bench.other_config("C5", torch.device("cuda", 0), steps=600, warmup=200)
print("FUSE_BLOCKS", os.environ["MOPOE_FUSE_BLOCKS"], "C5 %.2f us" % (1e3 * r["ms_per_step"]), r["roofline"]["kernels_avg_us"], flush=True)
PY
done 2>&1 | grep -v amdgpu.ids
