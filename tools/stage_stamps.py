#!/usr/bin/env python3
"""Diagnostic: where k_latent spends its time.  Uses the -DMOPOE_STAMPS build
(make -C 2022_cambroise_interpret_multivae_amd/csrc stamps):
    MOPOE_LIB=libmopoe_hip_stamps.so python tools/stage_stamps.py
Prints per-stage durations (100 MHz s_memrealtime) of block 0 and the shader
clock it ran at (delta s_memtime / delta s_memrealtime x 100 MHz)."""
import os
import sys

os.environ.setdefault("MOPOE_LIB", "libmopoe_hip_stamps.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import mopoe_amd as mm  # noqa: E402

spec = mm.ModelSpec(["clinical", "rois"], [7, 444], [3, 20])
eng = mm.MoPoEEngine(spec, "cuda", seed=1)
g = torch.Generator().manual_seed(0)
x = {"clinical": torch.randn(256, 7, generator=g).cuda(),
     "rois": torch.randn(256, 444, generator=g).cuda()}
names = ["S0 h->LDS", "S1 heads", "S2a combine+x", "S2b fusion", "S2c KL sums",
         "S3 decoder+nll", "S4 g_z", "S4b sum (+pass end)", "S5 bwd fusion", "S6 g_pre",
         "final partials"]
for it in range(300):
    plan, ws = eng.train_step(x)
torch.cuda.synchronize()
acc = None
for it in range(20):
    plan, ws = eng.train_step(x)
    torch.cuda.synchronize()
    st = ws._stats_all[64:].cpu().view(torch.int64)[:24].view(12, 2)
    d = (st[1:] - st[:-1]).double()
    acc = d if acc is None else acc + d
acc /= 20
tot = acc[:, 0].sum().item()
for nme, row in zip(names, acc):
    rt, mt = row[0].item(), row[1].item()
    print("%-20s %7.2f us   clock %6.0f MHz" % (nme, rt / 100.0,
                                                 (mt / rt * 100.0) if rt else 0))
print("%-20s %7.2f us" % ("total block 0", tot / 100.0))
