"""checkpoints in the reference's layout, readable by both
sides, and the reload of a trained run (MultimodalExperiment.get_experiment,
reference multimodal_cohort/experiment.py:93-121).

CPU part (needs no GPU: parameters live in a host buffer): layout, the checkpoint the
reference would pick, the weights-only loaders -- and, in the development container
where /root/reference exists, the reference's own model loading our file and our
model loading the reference's."""
import os
import types
from importlib import import_module

import pytest
import torch

import mopoe_oracle as mo
from surface_util import make_flags

_P = "2022_cambroise_interpret_multivae_amd."
checkpoint = import_module(_P + "checkpoint")
experiment = import_module(_P + "multimodal_cohort.experiment")


def _flags(tmp, device="cpu", method="joint_elbo", **topo):
    cfg = mo.Config(["clinical", "rois"], [7, 444], [3, 20], method=method, **topo)
    flags = make_flags(cfg, device)
    flags.dir_checkpoints = os.path.join(tmp, "checkpoints")
    flags.dir_experiment_run = tmp
    os.makedirs(flags.dir_checkpoints)
    return cfg, flags


def test_layout_pick_rule_and_safe_loaders(tmp_path):
    cfg, flags = _flags(str(tmp_path))
    exp = experiment.MultimodalExperiment(flags)
    assert list(exp.modalities) == ["clinical", "rois"]
    assert list(exp.subsets) == ["", "clinical", "rois", "clinical_rois"]
    model = exp.models
    saved = {}
    for epoch in (4, 9, 11):
        with torch.no_grad():
            model.engine.params.add_(0.25)
        path = checkpoint.save_model(model, flags, epoch)
        assert path == os.path.join(flags.dir_checkpoints, "%04d" % epoch, "model")
        saved[epoch] = {k: v.clone() for k, v in model.state_dict().items()}
    for name in ("enc_clinical", "dec_clinical", "enc_rois", "dec_rois"):
        sd = checkpoint.load_state(os.path.join(flags.dir_checkpoints, name), "cpu")
        assert all(isinstance(v, torch.Tensor) for v in sd.values())
    flags_file = os.path.join(str(tmp_path), "flags.rar")
    checkpoint.save_flags(flags, flags_file)
    back = checkpoint.load_flags(flags_file)           # weights-only loader
    assert isinstance(back, types.SimpleNamespace) and back.style_dim == [3, 20]
    files = checkpoint.find_checkpoints(flags.dir_checkpoints, "model")
    assert [int(f.split(os.sep)[-2]) for f in files] == [4, 9, 11]
    # the reference's rule (experiment.py:113-117): latest, or argmin(epochs >= load_epoch)
    import numpy as np
    for load_epoch in (None, 0, 5, 9, 10, 12, 99):
        want = files[-1] if load_epoch is None else files[int(np.argmin(
            np.array([4, 9, 11]) >= load_epoch))]
        assert checkpoint.pick_checkpoint(files, load_epoch) == want
    with pytest.raises(ValueError):
        checkpoint.pick_checkpoint([])
    if not torch.cuda.is_available():                  # get_experiment puts the model on
        exp2, f2 = experiment.MultimodalExperiment.get_experiment(   # cuda when there is one
            flags_file, flags.dir_checkpoints)
        same = lambda m, sd: all(torch.equal(v, sd[k]) for k, v in m.state_dict().items())
        assert same(exp2.models, saved[11])
        exp3, _ = experiment.MultimodalExperiment.get_experiment(
            flags_file, flags.dir_checkpoints, load_epoch=10)
        assert same(exp3.models, saved[4])      # (argmin picks the oldest here)


@pytest.mark.reference
@pytest.mark.parametrize("topo", [{}, dict(enc_layers=2, dec_layers=1, dropout=0.2),
                                  dict(enc_layers=0, dec_layers=2, sample_scale=True)],
                         ids=["default", "enc2_dec1_drop", "enc0_dec2_sample_scale"])
def test_reference_model_and_ours_read_each_others_checkpoints(tmp_path, topo):
    import ref_harness as rh
    if not rh.reference_available():
        pytest.skip("needs /root/reference (development container)")
    ns = rh.import_reference()
    cfg, flags = _flags(str(tmp_path), **topo)
    ours = experiment.MultimodalExperiment(flags).models
    ours.load_state_dict(mo.init_params(cfg, 3))
    path = checkpoint.save_model(ours, flags, 7)
    rflags = rh.make_flags(cfg.input_dim, cfg.style_dim, **topo)
    ref = rh.build_experiment(ns, rflags, cfg.names).models
    missing, unexpected = ref.load_state_dict(torch.load(path, weights_only=True), strict=True)
    assert not missing and not unexpected
    for k, v in ref.state_dict().items():
        assert torch.equal(v, ours.state_dict()[k]), k
    # per-modality files, as the reference's Modality.save_networks names them
    enc = torch.load(os.path.join(flags.dir_checkpoints, "enc_rois"), weights_only=True)
    ref.encoders["rois"].load_state_dict(enc, strict=True)
    # the other direction: a state dict written by the reference model
    for p in ref.parameters():
        torch.nn.init.normal_(p, std=0.1)
    rpath = os.path.join(flags.dir_checkpoints, "0012")
    os.makedirs(rpath)
    torch.save(ref.state_dict(), os.path.join(rpath, "model"))
    flags_file = os.path.join(str(tmp_path), "flags.rar")
    checkpoint.save_flags(flags, flags_file)
    if not torch.cuda.is_available():
        exp2, _ = experiment.MultimodalExperiment.get_experiment(flags_file, flags.dir_checkpoints)
        for k, v in ref.state_dict().items():
            assert torch.equal(v, exp2.models.state_dict()[k]), k
