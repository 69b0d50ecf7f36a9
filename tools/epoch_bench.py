#!/usr/bin/env python3
"""Throughput of the reference-shaped training loop (run_epochs.train of the mirror
package), i.e. what a user who swaps the modules gets, next to bench.py's bare
engine loop: (a) epochs over a ResidentCohort (index batches, f1), (b) the reference's
zero_grad / backward / step sequence over ready device batches."""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [root, os.path.join(root, "tests"), os.path.join(root, "oracle")]
import numpy as np
import torch
from importlib import import_module
import mopoe_oracle as mo
from surface_util import make_experiment, run_epochs
ds_mod = import_module("2022_cambroise_interpret_multivae_amd.multimodal_cohort.dataset")

n, bs = 256 * 64, 256
rng = np.random.RandomState(0)
data = {"clinical": rng.randn(n, 7), "rois": rng.randn(n, 444)}
idx = {m: np.array(list(range(n)), dtype=object) for m in data}
ds = ds_mod.MultimodalDataset(data, idx)
cfg = mo.Config(["clinical", "rois"], [7, 444], [3, 20])
exp = make_experiment(cfg, "cuda")
exp.flags.batch_size = bs
exp.dataset_train = ds_mod.ResidentCohort(ds, "cuda")
for _ in range(3):
    run_epochs.train(0, 0, exp, None)
torch.cuda.synchronize()
t0 = time.perf_counter()
E = 10
for _ in range(E):
    run_epochs.train(0, 0, exp, None)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("(a) ResidentCohort epochs : %.1f us/step, %.2f M samples/s" % (1e6 * dt / (E * n / bs), E * n / dt / 1e6))

batches = [({"clinical": torch.randn(bs, 7).cuda(), "rois": torch.randn(bs, 444).cuda()}, None, None)
           for _ in range(64)]
exp2 = make_experiment(cfg, "cuda")
exp2.flags.batch_size = bs
exp2.dataset_train = iter(())
class Ready:
    def __iter__(self):
        return iter(batches)
exp2.dataset_train = Ready()
for _ in range(3):
    run_epochs.train(0, 0, exp2, None)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(E):
    run_epochs.train(0, 0, exp2, None)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("(b) zero_grad/backward/step: %.1f us/step, %.2f M samples/s" % (1e6 * dt / (E * 64), E * 64 * bs / dt / 1e6))
