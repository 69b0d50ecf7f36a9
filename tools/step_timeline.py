#!/usr/bin/env python3
"""Diagnostic (stamps build): one training step on the device's 100 MHz realtime counter --
entry/exit of one block of each of the three kernels, so the gaps between the launches and
the phases inside k_wgrad can be read off.  MOPOE_LIB must point at libmopoe_hip_stamps.so."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mopoe_amd as mm
import bench
cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C1"]     # C1 | C3 | C5 | N64K
n = int(sys.argv[2]) if len(sys.argv) > 2 else cfg["batch"]
spec = bench.make_spec(cfg)
eng = mm.MoPoEEngine(spec, "cuda", seed=1)
g = torch.Generator().manual_seed(0)
pool = [{k: torch.randn(n, d, generator=g).cuda() for k, d in zip(cfg["names"], cfg["dims"])} for _ in range(8)]
print("step timeline, %s, %d rows" % (cfg["label"], n))
for i in range(300): eng.train_step(pool[i % 8])
torch.cuda.synchronize()
names = ["k_linear entry", "k_linear row index staged", "k_linear x tile staged", "k_linear MFMA done", "k_linear exit", "k_latent entry", "k_latent stamp0", "k_latent last stage end", "k_latent exit",
         "k_wgrad entry", "k_wgrad decoded, Adam operands requested", "k_wgrad GEMM done", "k_wgrad block reduced", "k_wgrad exit",
         "k_wgrad finalize block entry", "k_wgrad finalize block exit", "k_wgrad last logvar block entry", "k_wgrad last logvar block exit"]
acc = None
prev = None
rows = []
for it in range(60):
    plan, ws = eng.train_step(pool[it % 8]); torch.cuda.synchronize()
    c = eng.counters.cpu().view(torch.int32)
    s = ws._stats_all.cpu().view(torch.int32)
    lat = s[128:128 + 32].view(16, 2)
    cur = [int(c[64 + 11]), int(c[64 + 13]), int(c[64 + 14]), int(c[64 + 15]), int(c[64 + 12]), int(s[128 + 45]), int(lat[0, 0]), int(lat[10, 0]), int(s[128 + 46]),
           int(s[128 + 40]), int(s[128 + 41]), int(s[128 + 42]), int(s[128 + 43]), int(s[128 + 44]),
           int(s[128 + 60]), int(s[128 + 61]), int(s[128 + 62]), int(s[128 + 63])]
    rows.append(cur)
import numpy as np
r = np.array(rows, dtype=np.int64) & 0xFFFFFFFF
rel = ((r - r[:, :1]) % (1 << 32)) / 100.0
med = np.median(rel, axis=0)
for nm, v in zip(names, med):
    print("%-45s %7.2f us" % (nm, v))
