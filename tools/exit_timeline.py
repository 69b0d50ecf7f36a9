#!/usr/bin/env python3
"""Diagnostic: the row group's timeline inside the fused launch WITHOUT in-kernel stamps.
In the -DMOPOE_KNOCK build (make -C .../csrc knock) bits 24..27 of MOPOE_KNOCK make every
row group leave at stage boundary k: the launch, timed with HIP events, then lasts as long
as the groups need to get there.  Three timelines: the full kernel; the row groups alone
(producers leave at once, nobody waits for them); the skeleton (every phase left out but
the argument prefetch).
    python tools/exit_timeline.py [C1|C3|C5]"""
import os
import sys

os.environ.setdefault("MOPOE_LIB", "libmopoe_hip_knock.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import mopoe_amd as mm  # noqa: E402

import bench  # noqa: E402

cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C1"]
spec = bench.make_spec(cfg)
g = torch.Generator().manual_seed(0)
pool = [{k: torch.randn(cfg["batch"], d, generator=g).cuda() for k, d in zip(cfg["names"], cfg["dims"])}
        for _ in range(16)]
STEPS = int(os.environ.get("KNOCK_STEPS", "1000"))
POINTS = [(12, "entry (argument prefetch issued)"), (0, "S0: requests out, LDS zeroed"),
          (1, "S0 end: h in LDS"), (13, "S1: K parts of the heads in LDS"), (2, "S1 end: heads"), (3, "(x tiles, late form)"),
          (4, "S2b end: fusion forward"), (6, "S3 end: decoder + NLL"), (7, "S4 end: dL/dz"),
          (8, "decoder passes done"), (9, "S5 end: fusion backward"), (14, "S6: K parts of dL/dh in LDS"), (10, "S6 end: dL/dh"),
          (11, "tail")]


def run(mask):
    os.environ["MOPOE_KNOCK"] = str(mask)
    mm._lib.reload_knobs()
    eng = mm.MoPoEEngine(spec, "cuda", seed=1)
    eng.check_valid = lambda sync=False: None
    for i in range(150):
        eng.train_step(pool[i % 16])
    torch.cuda.synchronize()
    mm._lib.profile_enable(True)
    for i in range(STEPS):
        eng.train_step(pool[i % 16])
    torch.cuda.synchronize()
    prof = mm._lib.profile_read()
    mm._lib.profile_enable(False)
    return prof["k_fused"][1] / prof["k_fused"][0] * 1e3


ALL = (1 << 13) - 1          # every phase but the argument prefetch (bit 13)
base = {"full kernel": 0, "row groups alone": 1 << 14 | 1, "skeleton of a row group": 1 << 14 | ALL}
print("k_fused (HIP events, incl. ~2.3 us of event overhead) when every row group leaves at ...")
print("%-36s %12s %18s %12s" % ("", *base))
bare = run(1 << 14 | 1 << 15)
print("%-36s %12.2f" % ("bare launch (all leave at once)", bare))
for k, label in POINTS:
    row = [run(m | (k + 1) << 24) for m in base.values()]
    print("%-36s %12.2f %18.2f %12.2f" % (label, *row), flush=True)
row = [run(m) for m in base.values()]
print("%-36s %12.2f %18.2f %12.2f" % ("(no early exit)", *row))
