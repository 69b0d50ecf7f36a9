#!/usr/bin/env python3
"""Diagnostic: host (enqueue) time per training step of a general topology's chain of launches
next to the device time, configs[1]'s shapes: when the first exceeds the second the step is the
host's."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

mm = bench.mm
c = bench.CONFIGS[os.environ.get("TOPOLOGY_CONFIGS", "C1")]
for label, kw in (("enc 2, dec 1, dropout 0.2", dict(enc_layers=2, dec_layers=1, dropout=0.2)),
                  ("enc 2, dec 1", dict(enc_layers=2, dec_layers=1)),
                  ("default", {})):
    spec = mm.ModelSpec(c["names"], c["dims"], c["style"], class_dim=bench.LATENT, method=c["method"], **kw)
    eng = mm.MoPoEEngine(spec, "cuda", seed=1)
    eng.reset_parameters(torch.Generator().manual_seed(0))
    pool = bench.make_pool(c, torch.device("cuda"), count=8)
    for i in range(300):
        eng.train_step(pool[i % 8])
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        for i in range(60):            # (short: the device queue never fills, the host never waits)
            eng.train_step(pool[i % 8])
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for i in range(1500):
            eng.train_step(pool[i % 8])
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        print("%-28s host %6.1f us/step to enqueue 60 steps (%6.1f until drained) | 1,500 steps: %6.1f us/step"
              % (label, 1e6 * (t1 - t0) / 60, 1e6 * (t2 - t0) / 60, 1e6 * (t3 - t2) / 1500), flush=True)
