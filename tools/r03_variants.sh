# times bench.py's configurations under every libmopoe_hip_v*.so experiment build
for lib in libmopoe_hip.so $(cd 2022_cambroise_interpret_multivae_amd && ls libmopoe_hip_v*.so); do
MOPOE_LIB=$lib python - <<'PY'
import os, sys, torch
sys.path.insert(0, ".")
import bench
dev = torch.device("cuda", 0)
out = []
for k in os.environ.get("VARIANT_CONFIGS", "C5,C3,C1").split(","):
    r = bench.other_config(k, dev, steps=600, warmup=200)
    out.append("%s %.2f us (fused %.2f, wgrad %.2f)" % (k, 1e3 * r["ms_per_step"], r["roofline"]["kernels_avg_us"]["k_fused"], r["roofline"]["kernels_avg_us"]["k_wgrad"]))
print(os.environ["MOPOE_LIB"], " | ".join(out), flush=True)
PY
done
