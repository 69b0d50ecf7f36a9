#!/usr/bin/env python3
"""f2 data point: the daa workflow's M stochastic forwards of a 50-row batch
(reference workflow.py:388-396, M = 1000) as ONE folded launch vs the loop of M
launches, and the 150 x 7 perturbed forwards (workflow.py:405-419) likewise."""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import torch
from importlib import import_module
import mopoe_amd as mm
from mopoe_amd import _lib as L
daa = import_module("2022_cambroise_interpret_multivae_amd.daa")
import mopoe_oracle as mo
from surface_util import make_experiment

cfg = mo.Config(["clinical", "rois"], [7, 444], [3, 20], method="joint_elbo")
model = make_experiment(cfg, "cuda").models
n, M, n_samples = 50, 1000, 150
g = torch.Generator().manual_seed(0)
data = {"clinical": torch.randn(n, 7, generator=g).cuda(), "rois": torch.randn(n, 444, generator=g).cuda()}

def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

def loop():
    for _ in range(M):
        model(data, sample_latents=True)["rec"]["rois"].loc
t_loop = timed(loop, 2)
t_fold = timed(lambda: daa.repeated_reconstructions(model, data, M), 20)
print("M=%d stochastic forwards of %d rows: loop %.2f ms, folded %.3f ms (%.0fx), %.1f M rows/s"
      % (M, n, t_loop * 1e3, t_fold * 1e3, t_loop / t_fold, M * n / t_fold / 1e6))
L.profile_enable(True)
daa.repeated_reconstructions(model, data, M); torch.cuda.synchronize()
for name, (cnt, ms) in L.profile_read().items():
    if cnt:
        print("   %-10s %8.1f us" % (name, 1e3 * ms / cnt))
L.profile_enable(False)
sv = torch.randn(n_samples, n, 7, generator=g).cuda()
t_fold2 = timed(lambda: daa.perturbed_reconstructions(model, data, sv, "likelihood", True), 20)
def loop2():
    for s in range(n_samples):
        for i in range(7):
            c = data["clinical"].clone(); c[:, i] = sv[s, :, i]
            model({"clinical": c, "rois": data["rois"]}, sample_latents=True)["rec"]["rois"].loc
t_loop2 = timed(loop2, 2)
print("%d x 7 perturbed forwards: loop %.2f ms, folded %.3f ms (%.0fx)"
      % (n_samples, t_loop2 * 1e3, t_fold2 * 1e3, t_loop2 / t_fold2))
