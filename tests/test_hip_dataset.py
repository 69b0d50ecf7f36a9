"""GPU, SURVEY.md section 8f row f1: index batches over HBM-resident blocks (row_index
of the C ABI).  Every batch of a MissingModalitySampler epoch is checked against the
ORACLE run on the gathered, scaled rows (loss terms, posteriors, gradients), and
against the materialised-batch HIP path bit for bit; the scaler is the reference's
StandardScaler (tests/golden/scaler.npz pins the CPU side); a full sampler epoch of
run_epochs.train."""
from collections import OrderedDict
from importlib import import_module

import numpy as np
import pytest
import torch

import mopoe_oracle as mo
from hip_util import Report, TOL, compare_forward, make_engine
from surface_util import make_experiment, run_epochs

pytestmark = pytest.mark.gpu
ds_mod = import_module("2022_cambroise_interpret_multivae_amd.multimodal_cohort.dataset")


def synthetic_cohort(n=300, seed=0):
    rng = np.random.RandomState(seed)
    has = {"clinical": rng.rand(n) > 0.2, "rois": rng.rand(n) > 0.2}
    has["clinical"] |= ~has["rois"]
    data, idx_per_mod = {}, {}
    for mod, d in (("clinical", 7), ("rois", 444)):
        rows = np.flatnonzero(has[mod])
        perm = rng.permutation(len(rows))
        data[mod] = rng.randn(len(rows), d) * 3.0 + 1.5
        col = np.empty(n, dtype=object)
        col[:] = None
        for k, subj in enumerate(rows):
            col[subj] = int(perm[k])
        idx_per_mod[mod] = col
    return ds_mod.MultimodalDataset(data, idx_per_mod)


def test_index_batches_equal_materialised_batches():
    ds = synthetic_cohort()
    from sklearn.preprocessing import StandardScaler
    scalers, scaled = {}, {}
    for mod in ds.modalities:
        sc = StandardScaler().fit(ds.data[mod])
        scalers[mod] = (sc.mean_, sc.scale_)
        scaled[mod] = sc.transform(ds.data[mod]).astype(np.float32)
    cohort = ds_mod.ResidentCohort(ds, "cuda", scalers=scalers)
    for mod in ds.modalities:       # scaled once on the way in == per-sample transform
        assert np.allclose(cohort.x[mod].cpu().numpy(), scaled[mod], rtol=1e-6, atol=1e-6)
    cfg = mo.Config(["clinical", "rois"], [7, 444], [3, 20])
    _, eng_a = make_engine(cfg)
    _, eng_b = make_engine(cfg)
    np.random.seed(4)
    batches = list(ds_mod.MissingModalitySampler(ds, 48))
    seen = set()
    spec = eng_a.spec
    rep = Report("gather batches vs oracle")
    for k, b in enumerate(batches):
        inputs, row_index = cohort.batch(b)
        seen.add(tuple(inputs))
        gathered = OrderedDict((m, cohort.x[m][row_index[m].long().cuda()]) for m in inputs)
        # the oracle on the rows the reference's per-sample transform would deliver
        x_o = OrderedDict((m, torch.from_numpy(scaled[m][row_index[m].numpy()]))
                          for m in inputs)
        params = OrderedDict((n, v.cpu()) for n, v in eng_a.named_params().items())
        noise = mo.Noise(generator=mo.noise_rng(300 + k))
        out, grads = mo.loss_and_grads(params, cfg, x_o, noise)
        plan, ws_a = eng_a.train_step(inputs, row_index=row_index, eps=noise.tape)
        _, ws_b = eng_b.train_step(gathered, eps=noise.tape)
        torch.cuda.synchronize()
        assert torch.equal(ws_a.stats, ws_b.stats)
        p = "batch%d/" % k
        compare_forward(rep, spec, eng_a, plan, ws_a, out, prefix=p, check_scale=False)
        for n, g in grads.items():
            rep.close_scaled(p + "grad/" + n, eng_a.grad_views[n], g, TOL["grad"])
    rep.finish()
    assert torch.equal(eng_a.params, eng_b.params)            # bit-identical training
    assert seen == {("clinical",), ("rois",), ("clinical", "rois")}


@pytest.mark.parametrize("topo", [dict(enc_layers=2, dec_layers=1, dropout=0.2), dict(enc_layers=1, dec_layers=1, sample_scale=True)],
                         ids=["enc2-dec1-drop", "dec1-logvar-head"])
def test_index_batches_through_the_general_chain(topo):
    """The same over a general topology's chain of launches (its first layer and the likelihood in
    the output layer's epilogue read x through the row index): a step on index batches over the
    resident blocks equals the step on the gathered copy bit for bit -- gradients, scalars,
    parameters -- complete and missing-modality batches alike."""
    ds = synthetic_cohort()
    cohort = ds_mod.ResidentCohort(ds, "cuda")
    cfg = mo.Config(["clinical", "rois"], [7, 444], [3, 20], **topo)
    import mopoe_amd as mm
    spec, _ = make_engine(cfg)
    eng_a, eng_b = mm.MoPoEEngine(spec, "cuda", seed=11), mm.MoPoEEngine(spec, "cuda", seed=11)
    for e in (eng_a, eng_b):
        e.load_params(mo.init_params(cfg, 0))
    np.random.seed(5)
    seen = set()
    for k, b in enumerate(ds_mod.MissingModalitySampler(ds, 48)):
        inputs, row_index = cohort.batch(b)
        seen.add(tuple(inputs))
        gathered = OrderedDict((m, cohort.x[m][row_index[m].long().cuda()]) for m in inputs)
        _, ws_a = eng_a.train_step(inputs, row_index=row_index)     # (device-drawn noise and masks: same seed)
        _, ws_b = eng_b.train_step(gathered)
        torch.cuda.synchronize()
        assert torch.equal(ws_a.stats, ws_b.stats), k
        assert torch.equal(eng_a.grads, eng_b.grads), k
    eng_a.check_valid(sync=True)
    assert torch.equal(eng_a.params, eng_b.params)
    assert seen == {("clinical",), ("rois",), ("clinical", "rois")}


def test_train_epoch_over_resident_cohort():
    cfg = mo.Config(["clinical", "rois"], [7, 444], [3, 20])
    exp = make_experiment(cfg, "cuda")
    exp.flags.batch_size = 64
    ds = synthetic_cohort(200, seed=2)
    exp.dataset_train = ds_mod.ResidentCohort(ds, "cuda")
    before = exp.models.engine.params.clone()
    np.random.seed(0)
    run_epochs.train(0, 0, exp, None)
    torch.cuda.synchronize()
    n_batches = len(ds_mod.MissingModalitySampler(ds, 64))
    assert exp.models.engine.step_count() == n_batches
    assert not torch.equal(before, exp.models.engine.params)
    assert torch.isfinite(exp.models.engine.params).all()
    # the DataLoader route of the reference still works on the same dataset
    exp.dataset_train = ds
    exp.batch_sampler_cls = ds_mod.MissingModalitySampler
    np.random.seed(0)
    run_epochs.train(0, 1, exp, None)
    assert exp.models.engine.step_count() == 2 * n_batches
