#!/usr/bin/env python3
"""Diagnostic: where a long free-running run of a general topology on noise inputs leaves finite
numbers (tools/soak_chain.py: method poe, 2 + 1 hidden layers, dropout 0.2)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mopoe_amd as mm
L = mm._lib
method = sys.argv[1] if len(sys.argv) > 1 else "poe"
spec = mm.ModelSpec(["clinical", "rois"], [7, 444], [3, 20], method=method, enc_layers=2, dec_layers=1, dropout=0.2)
eng = mm.MoPoEEngine(spec, "cuda", seed=7)
eng.reset_parameters(torch.Generator().manual_seed(0))
g = torch.Generator().manual_seed(1)
pool = [{"clinical": torch.randn(256, 7, generator=g).cuda(),
         "rois": torch.randn(256, 444, generator=g).cuda()} for _ in range(16)]
for i in range(20000):
    plan, ws = eng.train_step(pool[i % 16])
    if i % 500 == 0 or i < 3:
        torch.cuda.synchronize()
        loss = float(ws.stats[L.STAT_TOTAL_LOSS])
        p = eng.params
        lv = [eng.views["decoders.%s.logvar" % n] for n in ("clinical", "rois")]
        print("step %5d loss %12.5g  max|p| %9.4g  logvar min %8.3f %8.3f  finite %s" % (
            i, loss, float(p.abs().max()), float(lv[0].min()), float(lv[1].min()), bool(torch.isfinite(p).all())), flush=True)
        if not torch.isfinite(p).all():
            break
