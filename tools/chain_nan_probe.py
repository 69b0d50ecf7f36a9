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
fine = int(sys.argv[2]) if len(sys.argv) > 2 else 10 ** 9     # from this step on: every 10 steps, with the heads
for i in range(20000):
    plan, ws = eng.train_step(pool[i % 16])
    if i >= fine and i % 10 == 0:
        torch.cuda.synchronize()
        hm = [float(ws.heads[m][:256].abs().max()) for m in range(2)]
        names = sorted(eng.views, key=lambda k: -float(eng.views[k].abs().max()) if torch.isfinite(eng.views[k]).all() else -1e30)[:2]
        print("step %5d loss %10.5g  max|heads| %9.4g %9.4g  largest: %s" % (
            i, float(ws.stats[L.STAT_TOTAL_LOSS]), hm[0], hm[1],
            ", ".join("%s %.3g" % (k, float(eng.views[k].abs().max())) for k in names)), flush=True)
        if not torch.isfinite(eng.params).all():
            bad = [k for k in eng.views if not torch.isfinite(eng.views[k]).all()]
            print("non-finite:", bad[:8], len(bad))
            break
    if i % 500 == 0 or i < 3:
        torch.cuda.synchronize()
        loss = float(ws.stats[L.STAT_TOTAL_LOSS])
        p = eng.params
        lv = [eng.views["decoders.%s.logvar" % n] for n in ("clinical", "rois")]
        print("step %5d loss %12.5g  max|p| %9.4g  logvar min %8.3f %8.3f  finite %s" % (
            i, loss, float(p.abs().max()), float(lv[0].min()), float(lv[1].min()), bool(torch.isfinite(p).all())), flush=True)
        if not torch.isfinite(p).all():
            break
