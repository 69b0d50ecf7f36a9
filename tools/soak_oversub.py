#!/usr/bin/env python3
"""Soak test of the four-row launch with more blocks than CUs (513..1,024 rows): N training steps
from the same state must end in bit-identical parameters run to run and no hand-off may time out.
Usage: python tools/soak_oversub.py [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import mopoe_amd as mm  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
for method, n in (("poe", 768), ("joint_elbo", 1024), ("poe", 1024)):
    res = []
    for rep in range(2):
        spec = mm.ModelSpec(["clinical", "rois"], [7, 444], [3, 20], method=method)
        eng = mm.MoPoEEngine(spec, "cuda", seed=7)
        eng.reset_parameters(torch.Generator().manual_seed(0))
        g = torch.Generator().manual_seed(1)
        pool = [{"clinical": torch.randn(n, 7, generator=g).cuda(),
                 "rois": torch.randn(n, 444, generator=g).cuda()} for _ in range(8)]
        for i in range(steps):
            eng.train_step(pool[i % 8])
        torch.cuda.synchronize()
        p = eng.params.double()
        res.append((eng.step_count(), float(p.sum()), float((p * p).sum()), int(eng.counters[2])))
    print(method, n, res[0], "bit-identical rerun" if res[0] == res[1] else "DIFFERENT: %r" % (res[1],),
          "no timeouts" if res[0][3] == 0 and res[1][3] == 0 else "TIMEOUTS", flush=True)
