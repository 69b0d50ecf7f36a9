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


def _cohort(n, seed):
    rng = np.random.RandomState(seed)
    has = {"clinical": rng.rand(n) > 0.15, "rois": rng.rand(n) > 0.25}
    has["clinical"] |= ~has["rois"]
    data, idx = {}, {}
    for mod, h in has.items():
        data[mod] = rng.randn(int(h.sum()), 3)
        col = np.empty(n, dtype=object)
        col[:] = None
        for k, s in enumerate(rng.permutation(np.flatnonzero(h))):
            col[s] = k
        idx[mod] = col
    return ds_mod.MultimodalDataset(data, idx)


@pytest.mark.parametrize("n,bs,seed", [(101, 16, 7), (640, 64, 1), (17, 1, 2), (5, 8, 3),
                                        (3000, 256, 4), (1, 4, 5), (2100, 2, 6)])
def test_c_sampler_draws_numpys_legacy_stream_bit_for_bit(n, bs, seed):
    """mopoe_sampler_epoch (MT19937 + random_interval + the legacy shuffle restated in C)
    against numpy itself: the same batches, and numpy's global generator left in the same
    state -- position included -- as after the reference's own np.random.choice calls."""
    ds = _cohort(n, seed)
    smp = ds_mod.MissingModalitySampler(ds, bs)
    np.random.seed(seed)
    np.random.rand(seed * 37 % 700)          # (start somewhere inside a block of 624)
    start = np.random.get_state()
    want = smp.draw_numpy(bs)
    after, probe = np.random.get_state(), np.random.rand(3)
    np.random.set_state(start)
    got = smp.draw(bs)
    assert len(got) == len(want) == len(smp)
    for g, w in zip(got, want):
        assert g.dtype == np.int64 and np.array_equal(g, w)
    mine = np.random.get_state()
    assert mine[2] == after[2] and np.array_equal(mine[1], after[1])
    assert np.array_equal(np.random.rand(3), probe)


def test_prefetched_epochs_are_the_epochs_of_the_sampler():
    """ResidentCohort.epoch_schedule draws the NEXT epoch ahead in a helper thread; what the
    loop sees -- batches AND numpy's global stream -- is what drawing at the moment of the
    call would have given, also when somebody else uses np.random in between."""
    ds = _cohort(700, 9)
    cohort = ds_mod.ResidentCohort(ds, "cpu")
    twin = ds_mod.MissingModalitySampler(ds, 64)

    def rows_of(schedule):
        return [{m: r.numpy().copy() for m, r in batch.row_index().items()}
                for batch, _, _ in schedule]

    np.random.seed(3)
    got = [rows_of(cohort.epoch_schedule(64)) for _ in range(2)]
    np.random.rand(5)                         # an outsider draws: the prefetched epoch is void
    got.append(rows_of(cohort.epoch_schedule(64)))
    got.append(rows_of(cohort.epoch_schedule(32)))   # another batch size: void as well
    tail = np.random.rand(2)
    np.random.seed(3)
    want = []
    for k, bs in enumerate((64, 64, 64, 32)):
        if k == 2:
            np.random.rand(5)
        epoch = []
        for b in twin.draw_numpy(bs):
            inputs, row_index = cohort.batch(b)
            epoch.append({m: r.numpy() for m, r in row_index.items()})
        want.append(epoch)
    assert np.array_equal(np.random.rand(2), tail)
    for g, w in zip(got, want):
        assert len(g) == len(w)
        for gb, wb in zip(g, w):
            assert gb.keys() == wb.keys()
            for m in gb:
                assert np.array_equal(gb[m], wb[m])
