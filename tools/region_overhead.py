#!/usr/bin/env python3
"""Diagnostic: what a timed region of K steps carries besides its K steps (the driver times 20):
synchronise - K x train_step - synchronise for several K, the intercept of the fit is the region's
own cost."""
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_INTERRUPT", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

mm = bench.mm
c = bench.CONFIGS["C1"]
eng = mm.MoPoEEngine(bench.make_spec(c), "cuda", seed=1)
eng.reset_parameters(torch.Generator().manual_seed(0))
pool = bench.make_pool(c, torch.device("cuda"), count=64)
for i in range(3000):
    eng.train_step(pool[i % 64])
torch.cuda.synchronize()
rows = []
for K in (1, 2, 5, 10, 20, 50, 100, 400):
    best = []
    for rep in range(30):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(K):
            eng.train_step(pool[i % 64])
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        best.append((t2 - t0, t1 - t0))
    best.sort()
    med = best[len(best) // 2]
    rows.append((K, med[0], med[1]))
    print("K=%4d  region %8.1f us (%.2f us/step)   host enqueue %7.1f us" % (K, 1e6 * med[0], 1e6 * med[0] / K, 1e6 * med[1]))
(k0, a0, _), (k1, a1, _) = rows[4], rows[-1]
slope = (a1 - a0) / (k1 - k0)
print("steady %.2f us/step; a region's own cost %.1f us" % (1e6 * slope, 1e6 * (a0 - slope * k0)))
