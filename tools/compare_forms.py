#!/usr/bin/env python3
"""Diagnostic: one training step through the fused launch and through the three-launch form
(MOPOE_NO_FUSE=1) from the same state -- which outputs differ, and by how much (they must not:
the library is built with -ffp-contract=off so that no instantiation gets its own choice of
fused multiply-adds)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mopoe_amd as mm
def run(nofuse):
    if nofuse: os.environ["MOPOE_NO_FUSE"] = "1"
    else: os.environ.pop("MOPOE_NO_FUSE", None)
    spec = mm.ModelSpec(["clinical", "rois"], [7, 444], [3, 20], method="joint_elbo")
    eng = mm.MoPoEEngine(spec, "cuda", seed=1)
    eng.reset_parameters(torch.Generator().manual_seed(0))
    g = torch.Generator().manual_seed(0)
    x = {"clinical": torch.randn(256, 7, generator=g).cuda(), "rois": torch.randn(256, 444, generator=g).cuda()}
    plan, ws = eng.train_step(x, apply_adam=False)
    torch.cuda.synchronize()
    out = {"heads0": ws.heads[0], "heads1": ws.heads[1], "sub_mu": ws.subsets_mu, "sub_lv": ws.subsets_logvar, "joint_mu": ws.joint_mu,
           "z0": ws.z[0], "z1": ws.z[1], "loc1": ws.loc[1], "g_xhat1": ws.g_xhat[1], "g_heads0": ws.g_heads[0], "g_heads1": ws.g_heads[1],
           "g_pre1": ws.g_pre[1], "stats": ws.stats, "grads": eng.grads}
    return {k: v.clone() for k, v in out.items()}
a, b = run(False), run(True)
for k in a:
    d = (a[k] - b[k]).abs().max().item()
    print("%-10s max|diff| %.3e %s" % (k, d, "" if d == 0 else "<-- differs"))
d = (a["g_heads1"] - b["g_heads1"]).abs()
print("g_heads1 differing elements per column:", (d > 0).sum(0).tolist())
d = (a["g_heads0"] - b["g_heads0"]).abs()
print("g_heads0 differing elements per column:", (d > 0).sum(0).tolist())
