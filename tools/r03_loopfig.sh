python - <<'PY'
import sys, torch
sys.path.insert(0, ".")
import bench
for rep in range(2):
    for n in (3000, 16384):
        r = bench.loop_figure(torch.device("cuda", 0), n)
        print(r["subjects"], r["us_per_step"], flush=True)
PY
