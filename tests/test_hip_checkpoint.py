"""GPU, SURVEY.md section 8f row f4: train, checkpoint in the reference's layout, reload
through MultimodalExperiment.get_experiment (reference experiment.py:93-121) and get
the identical model; generation helpers (reference BaseMMVae.py:242-312) against the
oracle's decoder on the same latents."""
import os
from collections import OrderedDict
from importlib import import_module

import numpy as np
import pytest
import torch

import mopoe_oracle as mo
from hip_util import Report, TOL
from surface_util import make_flags

pytestmark = pytest.mark.gpu
_P = "2022_cambroise_interpret_multivae_amd."
checkpoint = import_module(_P + "checkpoint")
experiment = import_module(_P + "multimodal_cohort.experiment")
run_epochs = import_module(_P + "run_epochs")
ds_mod = import_module(_P + "multimodal_cohort.dataset")
from test_hip_dataset import synthetic_cohort  # noqa: E402


class Batches:
    """An iterable of ready (inputs, labels, metadata) batches (no __getitem__: the
    loop takes it as it is instead of wrapping it in a DataLoader)."""

    def __init__(self, batches):
        self.batches = batches

    def __iter__(self):
        return iter([(dict(b[0]), b[1], b[2]) for b in self.batches])


def test_train_checkpoint_reload_round_trip(tmp_path):
    cfg = mo.Config(["clinical", "rois"], [7, 444], [3, 20])
    flags = make_flags(cfg, "cuda")
    flags.dir_checkpoints = os.path.join(str(tmp_path), "checkpoints")
    os.makedirs(flags.dir_checkpoints)
    flags.batch_size, flags.start_epoch, flags.end_epoch = 64, 0, 6
    train = ds_mod.ResidentCohort(synthetic_cohort(200, seed=2), "cuda")
    x_test = {k: v.cuda() for k, v in mo.make_inputs(cfg.names, cfg.input_dim, 40, 5).items()}
    exp = experiment.MultimodalExperiment(flags, dataset_train=train,
                                          dataset_test=Batches([(x_test, None, {})]))
    exp.set_optimizers()
    flags_file = os.path.join(str(tmp_path), "flags.rar")
    checkpoint.save_flags(flags, flags_file)
    np.random.seed(0)
    run_epochs.run_epochs(exp)
    torch.cuda.synchronize()
    # checkpoints every fifth epoch and after the last one (run_epochs.py:243-256)
    files = checkpoint.find_checkpoints(flags.dir_checkpoints, flags.model_save)
    assert [f.split(os.sep)[-2] for f in files] == ["0004", "0005"]
    again, flags2 = experiment.MultimodalExperiment.get_experiment(
        flags_file, flags.dir_checkpoints)
    assert flags2.style_dim == flags.style_dim and flags2.device.type == "cuda"
    assert torch.equal(again.models.engine.params[:exp.models.spec.c_model.off_ctrl],
                       exp.models.engine.params[:exp.models.spec.c_model.off_ctrl])
    with torch.no_grad():
        a = exp.models(x_test, sample_latents=False)
        b = again.models(x_test, sample_latents=False)
    for k in a["rec"]:
        assert torch.equal(a["rec"][k].loc, b["rec"][k].loc)
        assert torch.equal(a["rec"][k].scale, b["rec"][k].scale)
    assert torch.equal(a["latents"]["joint"][0], b["latents"]["joint"][0])
    # the epoch-4 file holds an earlier state
    older, _ = experiment.MultimodalExperiment.get_experiment(
        flags_file, flags.dir_checkpoints, load_epoch=5)
    assert not torch.equal(older.models.engine.params, exp.models.engine.params)


def test_generation_helpers_match_oracle_decoder():
    cfg = mo.Config(["clinical", "rois"], [7, 444], [3, 20])
    flags = make_flags(cfg, "cuda")
    exp = experiment.MultimodalExperiment(flags)
    model = exp.models
    params = mo.init_params(cfg, 4)
    model.load_state_dict(params)
    n = 33
    g = torch.Generator().manual_seed(1)
    content = torch.randn(n, cfg.class_dim, generator=g)
    styles = {name: torch.randn(n, cfg.style_dim[m], generator=g)
              for m, name in enumerate(cfg.names)}
    rep = Report("generation")
    lat = {"content": content.cuda(), "style": {k: v.cuda() for k, v in styles.items()}}
    means = model.generate_from_latents(lat)
    stats = model.generate_sufficient_statistics_from_latents(lat)
    for m, name in enumerate(cfg.names):
        loc, scale = mo.decoder_forward(params, cfg, m, styles[name], content)
        rep.close("mean/" + name, means[name], loc, *TOL["loc"])
        rep.close("loc/" + name, stats[name].loc, loc, *TOL["loc"])
        rep.close("scale/" + name, stats[name].scale, scale.expand_as(loc), 1e-6, 1e-7)
    # cond_generation: content = mu when the posterior is a point mass (logvar -> -inf is
    # not representable; logvar = -60 makes std 1e-13), styles are fresh prior draws
    x = mo.make_inputs(cfg.names, cfg.input_dim, n, seed=2)
    with torch.no_grad():
        post = model.inference({k: v.cuda() for k, v in x.items()})
    mu = post["subsets"]["clinical_rois"][0]
    cg = model.cond_generation({"clinical_rois": [mu, torch.full_like(mu, -60.0)]})
    assert list(cg) == ["clinical_rois"] and list(cg["clinical_rois"]) == cfg.names
    # without a style branch the generation is a deterministic function of the content
    cfg0 = mo.Config(["clinical", "rois"], [7, 444], [3, 20], factorized=False)
    exp0 = experiment.MultimodalExperiment(make_flags(cfg0, "cuda"))
    p0 = mo.init_params(cfg0, 5)
    exp0.models.load_state_dict(p0)
    with torch.no_grad():
        post0 = exp0.models.inference({k: v.cuda() for k, v in x.items()})
    mu0 = post0["joint"][0]
    cg0 = exp0.models.cond_generation({"j": [mu0, torch.full_like(mu0, -60.0)]})
    assert exp0.models.get_random_styles(4) == {"clinical": None, "rois": None}
    for m, name in enumerate(cfg0.names):
        loc, _ = mo.decoder_forward(p0, cfg0, m, None, mu0.cpu())
        rep.close("cond/" + name, cg0["j"][name], loc, *TOL["loc"])
    gen = model.generate(16)
    assert gen["rois"].shape == (16, 444) and torch.isfinite(gen["rois"]).all()
    assert abs(float(model._prior_draw(50000, 4).std()) - 1.0) < 0.02
    rep.finish()
