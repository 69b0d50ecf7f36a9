#!/usr/bin/env python3
"""Diagnostic: per-kernel HIP-event times for a few configurations."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import mopoe_amd as mm  # noqa: E402


def run(method, n, fused, names=("clinical", "rois"), dims=(7, 444), steps=300):
    spec = mm.ModelSpec(list(names), list(dims), [3, 20], method=method)
    eng = mm.MoPoEEngine(spec, "cuda", seed=1)
    g = torch.Generator().manual_seed(0)
    pool = [{nm: torch.randn(n, d, generator=g).cuda() for nm, d in zip(names, dims)}
            for _ in range(8)]
    for i in range(50):
        eng.train_step(pool[i % 8], apply_adam=fused)
        if not fused:
            eng.adam_step()
    torch.cuda.synchronize()
    mm._lib.profile_enable(True)
    for i in range(steps):
        eng.train_step(pool[i % 8], apply_adam=fused)
        if not fused:
            eng.adam_step()
    torch.cuda.synchronize()
    prof = mm._lib.profile_read()
    mm._lib.profile_enable(False)
    line = "  ".join("%s %.2f" % (k, v[1] / v[0] * 1e3) for k, v in prof.items() if v[0])
    print("%-10s N=%-6d fused=%d %s | %s us" % (method, n, fused, "x".join(map(str, dims)), line),
          flush=True)


if __name__ == "__main__" and "--fwd" not in sys.argv and "--none" not in sys.argv:
    run("joint_elbo", 256, True)
    run("joint_elbo", 256, False)
    run("joint_elbo", 1024, True)
    run("joint_elbo", 4096, True)
    run("joint_elbo", 65536, True, steps=30)
    run("poe", 1024, True)
    run("joint_elbo", 512, True, names=("a", "b", "c", "d"), dims=(7, 444, 128, 64))


def run_fwd(n=256, sample=True):
    spec = mm.ModelSpec(["clinical", "rois"], [7, 444], [3, 20])
    eng = mm.MoPoEEngine(spec, "cuda", seed=1)
    g = torch.Generator().manual_seed(0)
    pool = [{"clinical": torch.randn(n, 7, generator=g).cuda(),
             "rois": torch.randn(n, 444, generator=g).cuda()} for _ in range(8)]
    for i in range(50):
        eng.forward(pool[i % 8], sample=sample, fresh=False)
    torch.cuda.synchronize()
    mm._lib.profile_enable(True)
    for i in range(300):
        eng.forward(pool[i % 8], sample=sample, fresh=False)
    torch.cuda.synchronize()
    prof = mm._lib.profile_read()
    mm._lib.profile_enable(False)
    print("forward-only N=%d sample=%d | %s us" % (n, sample, "  ".join(
        "%s %.2f" % (k, v[1] / v[0] * 1e3) for k, v in prof.items() if v[0])), flush=True)


if __name__ == "__main__" and "--fwd" in sys.argv:
    run_fwd(256, True)
    run_fwd(256, False)
