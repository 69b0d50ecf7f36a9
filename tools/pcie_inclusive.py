#!/usr/bin/env python3
"""The PCIe-inclusive rate of configs[1]: the boundary of the C ABI takes
device pointers, so bench.py's `value` has its inputs resident in HBM.  The reference's loop gets
every batch as HOST float64 tensors from its DataLoader and moves / casts them per step
(run_epochs.py:85-86: `.to(device).float()`); this times the same step with that hand-over inside
the timed region -- pageable float64 (the reference's case) and pinned float32 host batches."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mopoe_amd as mm  # noqa: E402  (before torch: the package's host-wait policy)
import torch  # noqa: E402
import bench  # noqa: E402

c = bench.CONFIGS["C1"]
dev = torch.device("cuda")
eng = mm.MoPoEEngine(bench.make_spec(c), dev, seed=1)
eng.reset_parameters(torch.Generator().manual_seed(0))
g = torch.Generator().manual_seed(1234)
pool32 = [{n: torch.randn(c["batch"], d, generator=g) for n, d in zip(c["names"], c["dims"])} for _ in range(64)]
cases = {
    "resident in HBM (bench.py's value)": [{k: v.to(dev) for k, v in b.items()} for b in pool32],
    "host float64, pageable (the reference's DataLoader hand-over)": [{k: v.double() for k, v in b.items()} for b in pool32],
    "host float32, pinned": [{k: v.pin_memory() for k, v in b.items()} for b in pool32],
}
for label, pool in cases.items():
    def step(i):
        b = pool[i % 64]
        x = {k: v.to(dev, non_blocking=True).float() for k, v in b.items()}   # run_epochs.py:85-86
        eng.train_step(x)
    for i in range(300):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 2000
    for i in range(n):
        step(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("%-62s %7.1f us/step  %6.2f M samples/s" % (label, 1e6 * dt, c["batch"] / dt / 1e6), flush=True)
eng.check_valid(sync=True)
