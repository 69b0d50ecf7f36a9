#!/usr/bin/env python3
"""Diagnostic: where the host time of an epoch over a ResidentCohort goes (16 k synthetic
subjects, batch 256): dataset.epoch() and its sampler, engine.train_step's enqueue cost per
call, and a cProfile of the step loop."""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [root, os.path.join(root, "tests"), os.path.join(root, "oracle")]
import numpy as np, torch
from importlib import import_module
import mopoe_oracle as mo
from surface_util import make_experiment, run_epochs
ds_mod = import_module("2022_cambroise_interpret_multivae_amd.multimodal_cohort.dataset")
n, bs = 16384, 256
rng = np.random.RandomState(0)
data = {"clinical": rng.randn(n, 7), "rois": rng.randn(n, 444)}
idx = {m: np.array(list(range(n)), dtype=object) for m in data}
ds = ds_mod.MultimodalDataset(data, idx)
cfg = mo.Config(["clinical", "rois"], [7, 444], [3, 20])
exp = make_experiment(cfg, "cuda"); exp.flags.batch_size = bs
coh = ds_mod.ResidentCohort(ds, "cuda"); exp.dataset_train = coh
eng = exp.models.engine
for _ in range(2): run_epochs.train(0, 0, exp, None)
torch.cuda.synchronize()
t0 = time.perf_counter(); ep = list(coh.epoch(bs)); t1 = time.perf_counter()
print("dataset.epoch(): %.2f ms for %d batches = %.1f us/batch" % (1e3*(t1-t0), len(ep), 1e6*(t1-t0)/len(ep)))
t0 = time.perf_counter(); b = list(ds_mod.MissingModalitySampler(ds, bs)); t1 = time.perf_counter()
print("  of which the sampler: %.2f ms" % (1e3*(t1-t0)))
torch.cuda.synchronize()
t0 = time.perf_counter()
for inputs, ri in ep: eng.train_step(inputs, row_index=ri, apply_adam=True)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("65 x engine.train_step: host %.1f us/call (enqueue only), %.1f us/step until drained" % (1e6*(t1-t0)/len(ep), 1e6*(t2-t0)/len(ep)))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for inputs, ri in ep: eng.train_step(inputs, row_index=ri, apply_adam=True)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
