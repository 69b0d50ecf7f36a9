#!/usr/bin/env python3
"""SURVEY.md 8d, the second number: training throughput WITH the input side -- a synthetic
on-disk cohort in the reference's layout (`multiblock_idx_train.npz` of row-or-None per
subject, `{block}_data.npy`; fetchers/multiblock_fetcher.py:76-179), 3,000 subjects of
which 20 % lack one block, loaded with MultimodalDataset.from_files, scaled once and kept
resident in HBM (ResidentCohort), batches of 256 drawn by the MissingModalitySampler
mirror (homogeneous-subset batches, complete ones first), stepped by run_epochs.train.

    python tools/cohort_bench.py [subjects] [batch]"""
import os
import sys
import tempfile
import time

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [root, os.path.join(root, "tests"), os.path.join(root, "oracle")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
from importlib import import_module  # noqa: E402
import mopoe_oracle as mo  # noqa: E402
from surface_util import make_experiment, run_epochs  # noqa: E402

ds_mod = import_module("2022_cambroise_interpret_multivae_amd.multimodal_cohort.dataset")

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 256
rng = np.random.RandomState(0)
dims = {"clinical": 7, "rois": 444}
lacks = rng.rand(n) < 0.2                      # 20 % of the subjects lack ONE block
which = rng.rand(n) < 0.5
has = {"clinical": ~(lacks & which), "rois": ~(lacks & ~which)}
with tempfile.TemporaryDirectory() as d:
    idx = {}
    for mod, dim in dims.items():
        rows = np.flatnonzero(has[mod])
        np.save(os.path.join(d, mod + "_data.npy"), rng.randn(len(rows), dim) * 2.0 + 0.5)
        col = np.empty(n, dtype=object)
        col[:] = None
        for k, subj in enumerate(rows):
            col[subj] = k
        idx[mod] = col
    np.savez(os.path.join(d, "multiblock_idx_train.npz"), **idx)
    ds = ds_mod.MultimodalDataset.from_files(os.path.join(d, "multiblock_idx_train.npz"))
    scalers = {m: (np.asarray(ds.data[m]).mean(0), np.asarray(ds.data[m]).std(0)) for m in dims}
    cfg = mo.Config(list(dims), list(dims.values()), [3, 20])
    exp = make_experiment(cfg, "cuda")
    exp.flags.batch_size = bs
    t0 = time.perf_counter()
    exp.dataset_train = ds_mod.ResidentCohort(ds, "cuda", scalers=scalers)
    torch.cuda.synchronize()
    t_load = time.perf_counter() - t0
steps_per_epoch = len(list(ds_mod.MissingModalitySampler(ds, bs)))
for _ in range(3):
    run_epochs.train(0, 0, exp, None)
torch.cuda.synchronize()
E = 30
t0 = time.perf_counter()
for _ in range(E):
    run_epochs.train(0, 0, exp, None)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("cohort of %d subjects (%d with both blocks), batch %d: %d steps per epoch; blocks scaled + "
      "uploaded once in %.1f ms" % (n, int((has["clinical"] & has["rois"]).sum()), bs,
                                     steps_per_epoch, 1e3 * t_load))
print("run_epochs.train over the resident cohort: %.2f ms per epoch, %.1f us per step, "
      "%.2f M samples/s (sampler + index upload + steps; host-bound)"
      % (1e3 * dt / E, 1e6 * dt / (E * steps_per_epoch), E * n / dt / 1e6))
