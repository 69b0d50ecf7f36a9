#!/usr/bin/env python3
"""Experiment: configs[1]'s model at 65,536 rows through the general chain of launches
(MOPOE_FORCE_GENERAL=1: big GEMM launches + element-wise kernels) against the row-group
kernel k_latent -- which regime does a large batch belong to?"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import mopoe_amd as mm  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
spec = mm.ModelSpec(["clinical", "rois"], [7, 444], [3, 20], class_dim=20, method="joint_elbo")
print("general:", spec.general)
eng = mm.MoPoEEngine(spec, "cuda", seed=1)
eng.reset_parameters(torch.Generator().manual_seed(0))
g = torch.Generator().manual_seed(0)
pool = [{"clinical": torch.randn(n, 7, generator=g).cuda(), "rois": torch.randn(n, 444, generator=g).cuda()}
        for _ in range(2)]
for i in range(3):
    eng.train_step(pool[i % 2])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(10):
    eng.train_step(pool[i % 2])
torch.cuda.synchronize()
print("%d rows: %.1f us/step" % (n, 1e5 * (time.perf_counter() - t0)))
eng.check_valid(sync=True)
