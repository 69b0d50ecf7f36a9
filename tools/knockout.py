#!/usr/bin/env python3
"""Diagnostic: where the fused launch's time goes in a build WITHOUT in-kernel stamps (a
stamp reads the realtime counter through the scalar memory path and so waits for every
scalar load in flight: the stamped timeline serialises what normally overlaps).
Uses the -DMOPOE_KNOCK build (make -C .../csrc knock): MOPOE_KNOCK is a bit mask of
phases of a row group that are left out; the launch is timed with HIP events, phase by
phase, and the difference to the full launch is that phase's share of the critical
path.  Results of such launches are garbage (no update is applied); only time counts.
    python tools/knockout.py [N] [C1|C3|C5]      (bench.py's configurations)"""
import os
import sys

os.environ.setdefault("MOPOE_LIB", "libmopoe_hip_knock.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import mopoe_amd as mm  # noqa: E402

import bench  # noqa: E402  (the configurations)

cfg = bench.CONFIGS[sys.argv[2] if len(sys.argv) > 2 else "C1"]
n = int(sys.argv[1]) if len(sys.argv) > 1 and int(sys.argv[1]) > 0 else cfg["batch"]
spec = bench.make_spec(cfg)
g = torch.Generator().manual_seed(0)
pool = [{k: torch.randn(n, d, generator=g).cuda() for k, d in zip(cfg["names"], cfg["dims"])}
        for _ in range(16)]
STEPS = int(os.environ.get("KNOCK_STEPS", "1500"))
PHASES = ["wait for the producers", "noise", "x tiles", "h -> LDS", "S1 heads MFMA",
          "S2b fusion fwd", "S3 decoder units", "KL sums", "S4 g_z", "S5 fusion bwd",
          "S6 g_pre", "LDS zeroing", "heads W prefetch", "kernarg prefetch"]


def run(mask):
    os.environ["MOPOE_KNOCK"] = str(mask)
    mm._lib.reload_knobs()          # (the library reads its environment once)
    eng = mm.MoPoEEngine(spec, "cuda", seed=1)
    eng.check_valid = lambda sync=False: None
    for i in range(200):
        eng.train_step(pool[i % 16])
    torch.cuda.synchronize()
    mm._lib.profile_enable(True)
    for i in range(STEPS):
        eng.train_step(pool[i % 16])
    torch.cuda.synchronize()
    prof = mm._lib.profile_read()
    mm._lib.profile_enable(False)
    return prof["k_fused"][1] / prof["k_fused"][0] * 1e3


if os.environ.get("KNOCK_MASKS"):       # "label=mask;label=mask": just these, against the full launch
    full = run(0)
    print("k_fused, all phases: %.2f us" % full)
    for item in os.environ["KNOCK_MASKS"].split(";"):
        label, mask = item.rsplit("=", 1)
        t = run(int(mask, 0))
        print("  %-46s %6.2f us   (%+.2f)" % (label, t, t - full), flush=True)
    print("k_fused, all phases again: %.2f us" % run(0))
    sys.exit(0)
full = run(0)
print("k_fused, all phases (HIP events, incl. ~2 us of event overhead): %.2f us" % full)
ONLY = [int(v) for v in os.environ.get("KNOCK_ONLY", "").split(",") if v]   # phase numbers
for i, name in enumerate(PHASES):
    if os.environ.get("KNOCK_SHORT"):
        break
    if ONLY and i not in ONLY:
        continue
    t = run(1 << i)
    print("  without %-24s %6.2f us   (%+.2f)" % (name, t, t - full))
if ONLY:
    sys.exit(0)
allk = run((1 << len(PHASES)) - 1)
print("  without all of them             %6.2f us   (%+.2f)" % (allk, allk - full))
fwd = run(sum(1 << i for i in (4, 5, 6, 7, 8, 9, 10)))
print("  without S1..S6                  %6.2f us   (%+.2f)" % (fwd, fwd - full))
ALL = (1 << len(PHASES)) - 1
for label, mask in (("no producers (row groups in full)", 1 << 14 | 1),
                    ("no producers, row groups without all phases", 1 << 14 | ALL),
                    ("no row groups (producers in full)", 1 << 15),
                    ("neither (the bare launch of 224 workgroups)", 1 << 14 | 1 << 15)):
    t = run(mask)
    print("  %-46s %6.2f us   (%+.2f)" % (label, t, t - full))
if os.environ.get("MOPOE_QUAD") == "1":
    for label, mask in (("S4: no Wd reads", 1 << 17), ("S4: no MFMAs", 1 << 18),
                        ("S4: no partial stores", 1 << 19), ("S3: no global stores", 1 << 20),
                        ("S0: no Wd -> LDS requests", 1 << 16),
                        ("S4: scalar scaffolding only", 1 << 17 | 1 << 18 | 1 << 19),
                        ("S4 out", 1 << 8), ("S4 + 3 dependent scalar loads", 1 << 21)):
        t = run(mask)
        print("  %-46s %6.2f us   (%+.2f)" % (label, t, t - full))
again = run(0)
print("k_fused, all phases again: %.2f us" % again)
