"""CPU: the input side -- MultimodalDataset subset
bookkeeping and MissingModalitySampler against batches recorded from the
reference's sampler (tests/golden/sampler.npz, same numpy seed)."""
import json
from importlib import import_module

import numpy as np
import pytest

from golden_util import GOLDEN_DIR

ds_mod = import_module("2022_cambroise_interpret_multivae_amd.multimodal_cohort.dataset")


def make_dataset(z):
    has = {"clinical": z["has/clinical"], "rois": z["has/rois"]}
    n = len(has["clinical"])
    rng = np.random.RandomState(0)
    data, idx_per_mod = {}, {}
    for mod, d in (("clinical", 7), ("rois", 44)):
        rows = np.flatnonzero(has[mod])
        perm = rng.permutation(len(rows))              # block rows in another order
        data[mod] = rng.randn(len(rows), d)
        col = np.empty(n, dtype=object)
        col[:] = None
        for k, subj in enumerate(rows):
            col[subj] = int(perm[k])
        idx_per_mod[mod] = col
    return ds_mod.MultimodalDataset(data, idx_per_mod)


def test_sampler_reproduces_reference_batches():
    z = np.load(GOLDEN_DIR + "/sampler.npz", allow_pickle=False)
    ds = make_dataset(z)
    assert ds.modality_subsets == [("clinical",), ("rois",), ("clinical", "rois")]
    for seed, bs in ((7, 16), (11, 32), (5, 200)):
        want = json.loads(str(z["batches/%d_%d" % (seed, bs)]))
        np.random.seed(seed)
        smp = ds_mod.MissingModalitySampler(ds, batch_size=bs)
        got = [[int(i) for i in b] for b in smp]
        assert got == want
        assert len(smp) == int(z["len/%d_%d" % (seed, bs)]) == len(want)
        # every batch is homogeneous in its modality set; complete batches first
        sizes = [len(b) for b in got]
        first_small = next((i for i, s in enumerate(sizes) if s < bs), len(sizes))
        assert all(s < bs for s in sizes[first_small:])
        for b in got:
            sets = {tuple(m for m in ds.modalities if ds.idx_per_mod[m][i] is not None)
                    for i in b}
            assert len(sets) == 1
    assert sorted(i for b in got for i in b) == list(range(len(ds)))


def test_dataset_items_and_stratify_refusal():
    z = np.load(GOLDEN_DIR + "/sampler.npz", allow_pickle=False)
    ds = make_dataset(z)
    item, label, meta = ds[0]
    assert label == 0 and meta == {}
    for mod, v in item.items():
        assert np.array_equal(v.numpy(), ds.data[mod][int(ds.idx_per_mod[mod][0])])
    assert abs(sum(ds.get_modality_proportions()) - 1.0) < 1e-12
    with pytest.raises(NotImplementedError):
        ds_mod.MissingModalitySampler(ds, 8, stratify=["age"])


def test_scaler_reproduces_the_reference_transform_chain():
    """tests/golden/scaler.npz: the StandardScaler the reference's set_scalers fits
    (experiment.py:146-166) and the rows its per-sample transform chain delivers
    (experiment.py:228-232), recorded from the reference.  fit_scalers + ResidentCohort
    scale each block once, in float64, and round to float32 as the loop's .float() does."""
    z = np.load(GOLDEN_DIR + "/scaler.npz", allow_pickle=False)
    data, idx = {}, {}
    for mod in ("clinical", "rois"):
        data[mod] = z["data/" + mod]
        col = np.empty(len(z["idx/" + mod]), dtype=object)
        col[:] = [None if r < 0 else int(r) for r in z["idx/" + mod]]
        idx[mod] = col
    ds = ds_mod.MultimodalDataset(data, idx)
    scalers = ds_mod.fit_scalers(ds)
    for mod in ds.modalities:
        assert np.allclose(scalers[mod][0], z["mean/" + mod], rtol=1e-12, atol=1e-12)
        assert np.allclose(scalers[mod][1], z["scale/" + mod], rtol=1e-12, atol=1e-12)
    assert scalers["clinical"][1][3] == 1.0            # the constant feature
    cohort = ds_mod.ResidentCohort(ds, "cpu", scalers=scalers)
    for mod in ds.modalities:
        rows = [int(r) for r in z["idx/" + mod] if r >= 0]     # subjects that have it, in order
        got = cohort.x[mod][rows].numpy()
        want = z["transformed/" + mod]
        assert got.dtype == want.dtype == np.float32
        assert np.allclose(got, want, rtol=2e-7, atol=1e-7), np.abs(got - want).max()
