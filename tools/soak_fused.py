#!/usr/bin/env python3
"""Soak test of the fused launch's hand-off: N training steps from the same state must end in
bit-identical parameters run to run (four-row form) and with / without fusion (16-row form: a
stale read of h would show), and no hand-off may time out.
Usage: python tools/soak_fused.py [steps]"""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, root)
    import torch
    import mopoe_amd as mm
    steps = int(sys.argv[2])
    spec = mm.ModelSpec(["clinical", "rois"], [7, 444], [3, 20])
    eng = mm.MoPoEEngine(spec, "cuda", seed=7)
    eng.reset_parameters(torch.Generator().manual_seed(0))
    g = torch.Generator().manual_seed(1)
    pool = [{"clinical": torch.randn(256, 7, generator=g).cuda(),
             "rois": torch.randn(256, 444, generator=g).cuda()} for _ in range(16)]
    for i in range(steps):
        plan, ws = eng.train_step(pool[i % 16])
    torch.cuda.synchronize()
    p = eng.params.double()
    print("RESULT %d %.17g %.17g %d" % (eng.step_count(), float(p.sum()), float((p * p).sum()),
                                        int(eng.counters[2])))
    sys.exit(0)
steps = sys.argv[1] if len(sys.argv) > 1 else "20000"
out = {}
# the four-row form (default at this size) must repeat itself bit for bit; the 16-row form of
# the fused launch and the three launches are the same arithmetic in the same order
for name, env in (("four-row", {}), ("four-row again", {}), ("16-row fused", {"MOPOE_QUAD": "0"}),
                  ("three launches", {"MOPOE_QUAD": "0", "MOPOE_NO_FUSE": "1"})):
    e = dict(os.environ, **env)
    r = subprocess.run([sys.executable, __file__, "--child", steps], env=e, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
    if not line:
        sys.exit("child failed:\n" + r.stdout + r.stderr)
    out[name] = line[0]
    print("%-15s %s" % (name, line[0]), flush=True)
ok = out["four-row"] == out["four-row again"] and out["16-row fused"] == out["three launches"] and \
    all(v.split()[-1] == "0" for v in out.values())
print("bit-identical pairs, no timeouts" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
