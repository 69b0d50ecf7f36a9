#!/usr/bin/env python3
"""Diagnostic: the four-row form against the 16-row forms over batch sizes and methods
(MOPOE_QUAD, MOPOE_QUAD_MAX_N, MOPOE_FUSE_BLOCKS), same box: us per step and per kernel."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
import mopoe_amd as mm  # noqa: E402


def run(method, n, env):
    for k in ("MOPOE_QUAD", "MOPOE_QUAD_MAX_N", "MOPOE_FUSE_BLOCKS", "MOPOE_QUAD_MAX_N2", "MOPOE_QUAD_OVERSUB", "MOPOE_LIN_BIG_ROWS", "MOPOE_LIN_KS"):
        os.environ.pop(k, None)
    os.environ.update(env)
    mm._lib.reload_knobs()          # (the library reads its environment once)
    spec = mm.ModelSpec(["clinical", "rois"], [7, 444], [3, 20], method=method)
    eng = mm.MoPoEEngine(spec, "cuda", seed=1)
    g = torch.Generator().manual_seed(0)
    pool = [{"clinical": torch.randn(n, 7, generator=g).cuda(),
             "rois": torch.randn(n, 444, generator=g).cuda()} for _ in range(8)]
    for i in range(300):
        eng.train_step(pool[i % 8])
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for i in range(1000):
        eng.train_step(pool[i % 8])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 1000
    eng.check_valid(sync=True)
    mm._lib.profile_enable(True)
    for i in range(300):
        eng.train_step(pool[i % 8])
    torch.cuda.synchronize()
    prof = mm._lib.profile_read()
    mm._lib.profile_enable(False)
    plan = spec.plan(["clinical", "rois"], n, True, None, True, True)
    print("%-10s N=%-5d %-48s groups %3d  %7.2f us/step  %s" % (
        method, n, " ".join("%s=%s" % kv for kv in env.items()) or "(default)", plan.row_groups(),
        1e6 * dt, "  ".join("%s %.2f" % (k, v[1] / v[0] * 1e3) for k, v in prof.items() if v[0])),
        flush=True)


for rnd in range(2):
    for method, n in (("poe", 576), ("poe", 640), ("poe", 768), ("poe", 1024), ("joint_elbo", 576), ("joint_elbo", 640),
                      ("joint_elbo", 768), ("joint_elbo", 1024)):
        run(method, n, {"MOPOE_QUAD_MAX_N": "512"})                               # sixteen-row groups
        run(method, n, {"MOPOE_QUAD_MAX_N": "1024", "MOPOE_QUAD_OVERSUB": "0"})   # four-row groups: fit, or the encoder layer apart
        run(method, n, {"MOPOE_QUAD_MAX_N": "1024"})                              # four-row groups: more blocks than CUs
