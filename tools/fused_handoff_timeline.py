#!/usr/bin/env python3
"""Diagnostic (stamps build): the hand-off inside the fused launch on the device's 100 MHz
counter -- when each of the eight encoder-tile producers of row tile 0 has issued its
stores of h, when the row group is ready to wait, when it sees the flag, when h is in LDS."""
import os, sys
os.environ.setdefault("MOPOE_LIB", "libmopoe_hip_stamps.so")
sys.path.insert(0, "/root/repo")
import torch, numpy as np
import mopoe_amd as mm
n = 256
spec = mm.ModelSpec(["clinical", "rois"], [7, 444], [3, 20], method="joint_elbo")
eng = mm.MoPoEEngine(spec, "cuda", seed=1)
g = torch.Generator().manual_seed(0)
pool = [{"clinical": torch.randn(n, 7, generator=g).cuda(), "rois": torch.randn(n, 444, generator=g).cuda()} for _ in range(8)]
for i in range(300): eng.train_step(pool[i % 8])
torch.cuda.synchronize()
rows = []
for it in range(60):
    plan, ws = eng.train_step(pool[it % 8]); torch.cuda.synchronize()
    c = eng.counters.cpu().view(torch.int32)
    s = ws._stats_all.cpu().view(torch.int32)
    lat = s[128:128 + 32].view(16, 2)
    rows.append([int(s[128 + 45])] + [int(c[64 + 16 + i]) for i in range(8)] + [int(s[128 + 47]), int(s[128 + 48]), int(lat[1, 0])]
                + [int(c[64 + 32 + i]) for i in range(7)] + [int(s[128 + 50 + i]) for i in range(5)])
r = np.array(rows, dtype=np.int64) & 0xFFFFFFFF
rel = ((r - r[:, :1] + (1 << 31)) % (1 << 32) - (1 << 31)) / 100.0
med = np.median(rel, axis=0)
names = ["consumer entry"] + ["producer z%d cg%d stores issued" % (i // 4, i % 4) for i in range(8)] + ["consumer ready to wait", "consumer saw the flag", "S0 end"]
names += ["  producer z1 cg0: entry", "  x and W loads issued", "  x tile in LDS (barrier)", "  MFMAs done",
          "  K parts parked (barrier)", "  stores drained", "  flag raised"]
names += ["  consumer: before the x requests", "  x requested", "  noise drawn", "  pads zeroed", "  x parked"]
for nm, v in zip(names, med): print("%-34s %7.2f us" % (nm, v))
