#!/usr/bin/env python3
"""Diagnostic: the fused launch with more workgroups than CUs (MOPOE_FUSE_BLOCKS caps the
grid the 64- / 32-column encoder-layer blocks are used for; default 256 = all resident).
Same box, bench.py's configurations: us per step and per kernel (HIP events)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402

caps = [int(v) for v in os.environ.get("CAPS", "256,448,640,1024").split(",")]
for rnd in range(int(os.environ.get("ROUNDS", "2"))):
    for key in os.environ.get("CONFIGS", "C3,C5").split(","):
        for cap in caps:
            os.environ["MOPOE_FUSE_BLOCKS"] = str(cap)
            c = bench.CONFIGS[key]
            dt, eng, step, sp = bench.time_single_gpu(c, torch.device("cuda"), 1000, 300)
            prof = bench.profile_steps(step, 1300, 500)
            print("%s cap %-5d %7.2f us/step  %s" % (key, cap, 1e6 * dt / 1000, "  ".join(
                "%s %.2f" % (k, v[1] / v[0] * 1e3) for k, v in prof.items() if v[0])), flush=True)
