#!/usr/bin/env python3
"""Diagnostic: k_latent stage times as differences of whole-kernel times.
Uses the -DMOPOE_STAMPS build (make -C .../csrc stamps), in which
counters[15] = k makes k_latent return after stage k; each truncated kernel is
timed with HIP events (mopoe_profile_*), so nothing is instrumented inside."""
import os
import sys

os.environ.setdefault("MOPOE_LIB", "libmopoe_hip_stamps.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import mopoe_amd as mm  # noqa: E402

method = sys.argv[1] if len(sys.argv) > 1 else "joint_elbo"
spec = mm.ModelSpec(["clinical", "rois"], [7, 444], [3, 20], method=method)
eng = mm.MoPoEEngine(spec, "cuda", seed=1)
g = torch.Generator().manual_seed(0)
pool = [{"clinical": torch.randn(256, 7, generator=g).cuda(),
         "rois": torch.randn(256, 444, generator=g).cuda()} for _ in range(16)]
names = {1: "S0 h->LDS", 2: "S1 heads", 3: "S2a combine + x->LDS", 4: "S2b fusion fwd",
         5: "S2c KL sums", 6: "S3 decoder+NLL", 7: "S4 g_z partials", 8: "S4b sum",
         9: "S5 fusion bwd", 10: "S6 g_pre", 0: "tail (partials)"}
order = [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 0]
times = {}
for k in order:
    eng.counters[15] = k
    for i in range(100):
        eng.train_step(pool[i % 16], apply_adam=False)
    torch.cuda.synchronize()
    mm._lib.profile_enable(True)
    for i in range(300):
        eng.train_step(pool[i % 16], apply_adam=False)
    torch.cuda.synchronize()
    prof = mm._lib.profile_read()
    mm._lib.profile_enable(False)
    times[k] = prof["k_latent"][1] / prof["k_latent"][0] * 1e3
prev = 0.0
for k in order:
    print("%-24s %7.2f us   (kernel up to here %7.2f us)" % (names[k], times[k] - prev, times[k]))
    prev = times[k]
