#!/usr/bin/env python3
"""Diagnostic: the specialised instantiations of the fused launch against the generic one
(MOPOE_NO_LEAN=1), same box, per BASELINE configuration: us per training step and per
kernel (HIP events)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib.util  # noqa: E402

spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(
    os.path.dirname(os.path.abspath(__file__))), "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
import torch  # noqa: E402

for key in ("C1", "C3", "C5"):
    for label, env in (("specialised", None), ("generic (MOPOE_NO_LEAN)", "1")):
        if env:
            os.environ["MOPOE_NO_LEAN"] = env
        else:
            os.environ.pop("MOPOE_NO_LEAN", None)
        c = bench.CONFIGS[key]
        dt, eng, step, sp = bench.time_single_gpu(c, torch.device("cuda"), 1500, 300)
        prof = bench.profile_steps(step, 1800, 500)
        print("%s %-26s %7.2f us/step  %s" % (key, label, 1e6 * dt / 1500, "  ".join(
            "%s %.2f" % (k, v[1] / v[0] * 1e3) for k, v in prof.items() if v[0])))
