#!/usr/bin/env python3
"""Diagnostic: the drop-in loop over the 3,000-subject cohort of bench.py's `loop` figure,
batch by batch: what each kind of batch of an epoch (both blocks / one block, full / ragged)
costs when it is stepped alone, and what the epoch's own host work costs."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import types  # noqa: E402
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
from importlib import import_module  # noqa: E402

mm = bench.mm
P = "2022_cambroise_interpret_multivae_amd."
ds_mod = import_module(P + "multimodal_cohort.dataset")
run_epochs = import_module(P + "run_epochs")
device = torch.device("cuda")
subjects, batch = int(sys.argv[1]) if len(sys.argv) > 1 else 3000, 256
c = bench.CONFIGS["C1"]
rng = np.random.RandomState(0)
lacks, which = rng.rand(subjects) < 0.2, rng.rand(subjects) < 0.5
has = {"clinical": ~(lacks & which), "rois": ~(lacks & ~which)}
data, idx = {}, {}
for mod, dim in zip(c["names"], c["dims"]):
    rows = np.flatnonzero(has[mod])
    data[mod] = rng.randn(len(rows), dim)
    col = np.empty(subjects, dtype=object)
    col[:] = None
    for k, subj in enumerate(rows):
        col[subj] = k
    idx[mod] = col
ds = ds_mod.MultimodalDataset(data, idx)
cohort = ds_mod.ResidentCohort(ds, device, scalers=ds_mod.fit_scalers(ds))
eng = mm.MoPoEEngine(bench.make_spec(c), device, seed=7)
eng.reset_parameters(torch.Generator().manual_seed(0))
np.random.seed(1)
for _ in range(3):
    sched = cohort.epoch_schedule(batch)
kinds = {}
for inputs, ri, scale in sched:
    kinds.setdefault((tuple(sorted(inputs.keys())), inputs.n), inputs)
for (mods, n), inputs in sorted(kinds.items(), key=lambda kv: (-len(kv[0][0]), -kv[0][1])):
    for _ in range(200):
        eng.train_step(inputs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(1000):
        eng.train_step(inputs)
    torch.cuda.synchronize()
    count = sum(1 for i, _, _ in sched if tuple(sorted(i.keys())) == mods and i.n == n)
    print("%-22s n=%4d  x%2d per epoch  %6.2f us/step alone" % ("+".join(mods), n, count,
                                                              1e3 * (time.perf_counter() - t0)))
# the epoch as the loop runs it (order of the schedule), without the schedule's own cost
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    for inputs, ri, scale in sched:
        eng.train_step(inputs)
torch.cuda.synchronize()
print("the %d steps of one schedule, in order: %.2f us/step" % (len(sched), 1e6 * (time.perf_counter() - t0) / 50 / len(sched)))
t0 = time.perf_counter()
for _ in range(50):
    s2 = cohort.epoch_schedule(batch)
t1 = time.perf_counter()
print("epoch_schedule() alone: %.1f us per epoch (%d steps)" % (1e6 * (t1 - t0) / 50, len(s2)))
