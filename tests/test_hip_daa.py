"""f2: the repeated forwards of the daa workflow
(reference workflow.py:388-419) folded into the batch axis -- one launch must give
exactly what the separate forwards give, and match the oracle run once per repeat."""
import pytest
import torch

import mopoe_oracle as mo
from hip_util import Report, TOL, make_engine
from surface_util import make_experiment

pytestmark = pytest.mark.gpu


def _cfg(method="joint_elbo"):
    return mo.Config(["clinical", "rois"], [7, 444], [3, 20], method=method)


@pytest.mark.parametrize("method", ["joint_elbo", "moe"])
def test_folded_repeats_match_the_oracle_per_repeat(method):
    """n = 20 rows (3 mixture components: slices 6/6/8), 5 repeats with injected eps:
    every repeat must equal an oracle forward of the 20-row batch with its eps."""
    cfg = _cfg(method)
    spec, eng = make_engine(cfg)
    params = mo.init_params(cfg, 0)
    n, R = 20, 5
    g = torch.Generator().manual_seed(3)
    x = {"clinical": torch.randn(n, 7, generator=g), "rois": torch.randn(n, 444, generator=g)}
    eps = [torch.randn(R * n, 20, generator=g), torch.randn(R * n, 3, generator=g),
           torch.randn(R * n, 20, generator=g)]
    idx = torch.arange(n, dtype=torch.int32).repeat(R)
    plan, ws = eng.forward({k: v.cuda() for k, v in x.items()}, sample=True, eps=eps,
                           row_index=idx.cuda(), group_rows=n)
    torch.cuda.synchronize()
    rep = Report("folded repeats vs oracle, " + method)
    rt, at = TOL["loc"]
    lt, la = TOL["latent"]
    for r in range(R):
        tape = [e[r * n:(r + 1) * n] for e in eps]
        out = mo.forward(params, cfg, x, mo.Noise(tape=tape), sample_latents=True)
        sl = slice(r * n, (r + 1) * n)
        rep.close("rep%d/joint_mu" % r, ws.joint_mu[sl], out["latents"]["joint"][0], lt, la)
        rep.close("rep%d/joint_lv" % r, ws.joint_logvar[sl], out["latents"]["joint"][1], lt, la)
        for m, name in enumerate(cfg.names):
            rep.close("rep%d/loc/%s" % (r, name), ws.loc[m][sl], out["rec"][name][0], rt, at)
    rep.finish()


def test_daa_helpers_equal_the_separate_forwards():
    """sample_latents=False makes the forwards deterministic: the folded launch must
    reproduce, bit for bit, the loop of separate forwards the reference runs."""
    from importlib import import_module
    daa = import_module("2022_cambroise_interpret_multivae_amd.daa")
    exp = make_experiment(_cfg(), "cuda")
    model = exp.models
    model.load_state_dict(mo.init_params(_cfg(), 0))
    n, n_samples = 12, 4
    g = torch.Generator().manual_seed(5)
    data = {"clinical": torch.randn(n, 7, generator=g).cuda(),
            "rois": torch.randn(n, 444, generator=g).cuda()}
    rec = daa.repeated_reconstructions(model, data, 6, sample_latents=False)
    one = model(data, sample_latents=False)["rec"]
    for name in data:
        loc, scale = rec[name]
        assert loc.shape == (6, n, data[name].shape[1])
        for i in range(6):
            assert torch.equal(loc[i], one[name].loc), name
        assert torch.equal(scale.expand_as(one[name].loc), one[name].scale.expand_as(one[name].loc))
    # stochastic repeats differ from one another and average towards the mean path
    st = daa.repeated_reconstructions(model, data, 64, sample_latents=True)["rois"][0]
    assert not torch.equal(st[0], st[1])
    mean = daa.mean_reconstructions(model, data, 64)["rois"][0]   # (a fresh draw)
    assert mean.shape == (n, 444) and bool(torch.isfinite(mean).all())

    for strategy in ("likelihood", "linear"):
        shape = (n_samples, n, 7) if strategy == "likelihood" else (n, n_samples, 7)
        sv = torch.randn(*shape, generator=g).cuda()
        got = daa.perturbed_reconstructions(model, data, sv, strategy, sample_latents=False)
        assert got.shape == (n, 7, n_samples, 444)
        for sample_idx in range(n_samples):      # workflow.py:405-419, verbatim loop shape
            for idx in range(7):
                cdata = data["clinical"].clone()
                if strategy == "likelihood":
                    cdata[:, idx] = sv[sample_idx, :, idx]
                else:
                    cdata[:, idx] = sv[:, sample_idx, idx]
                want = model({"clinical": cdata, "rois": data["rois"]},
                             sample_latents=False)["rec"]["rois"].loc
                assert torch.equal(got[:, idx, sample_idx], want), (strategy, sample_idx, idx)


def test_group_rows_is_validated():
    cfg = _cfg()
    spec, eng = make_engine(cfg)
    x = {"clinical": torch.zeros(20, 7).cuda(), "rois": torch.zeros(20, 444).cuda()}
    with pytest.raises(ValueError):
        eng.forward(x, group_rows=7)


def test_folded_repeats_through_the_large_batch_encoder_kernel():
    """64 repeats of 40 rows = 2560 rows: the folded launch takes the 64x64-tile
    encoder kernel (with the row gather), a single 40-row forward the small-batch one.
    Different summation orders, same float32 tolerance."""
    from importlib import import_module
    daa = import_module("2022_cambroise_interpret_multivae_amd.daa")
    exp = make_experiment(_cfg(), "cuda")
    model = exp.models
    model.load_state_dict(mo.init_params(_cfg(), 0))
    n, M = 40, 64
    g = torch.Generator().manual_seed(9)
    data = {"clinical": torch.randn(n, 7, generator=g).cuda(),
            "rois": torch.randn(n, 444, generator=g).cuda()}
    rec = daa.repeated_reconstructions(model, data, M, sample_latents=False)
    one = model(data, sample_latents=False)["rec"]
    rep = Report("folded 2560 rows vs one forward")
    for name in data:
        for i in (0, 17, M - 1):
            rep.close("%s/rep%d" % (name, i), rec[name][0][i], one[name].loc, *TOL["loc"])
    rep.finish()


def test_daa_helpers_with_the_per_sample_scale_head():
    """learn_output_sample_scale (networks.py:57-59,73-75): the likelihood's scale is the
    output of a logvar HEAD, per sample and per forward -- the reference's loop reads
    `.scale` of every forward (workflow.py:388-398).  The folded launch returns it per repeat
    (the one combination of two built features that used to raise), equal bit for bit to
    the separate forwards, and equal to the oracle's."""
    from importlib import import_module
    daa = import_module("2022_cambroise_interpret_multivae_amd.daa")
    cfg = mo.Config(["clinical", "rois"], [7, 444], [3, 20], sample_scale=True)
    exp = make_experiment(cfg, "cuda")
    model = exp.models
    params = mo.init_params(cfg, 0)
    model.load_state_dict(params)
    n, M = 12, 5
    g = torch.Generator().manual_seed(15)
    data = {"clinical": torch.randn(n, 7, generator=g).cuda(),
            "rois": torch.randn(n, 444, generator=g).cuda()}
    rec = daa.repeated_reconstructions(model, data, M, sample_latents=False)
    one = model(data, sample_latents=False)["rec"]
    want = mo.forward(params, cfg, {k: v.cpu() for k, v in data.items()}, mo.Noise(tape=[]),
                      sample_latents=False)["rec"]
    rep = Report("DAA with the logvar head")
    for name in data:
        loc, scale = rec[name]
        assert loc.shape == scale.shape == (M, n, data[name].shape[1])
        for i in range(M):
            assert torch.equal(loc[i], one[name].loc) and torch.equal(scale[i], one[name].scale), name
        rep.close("loc/" + name, loc[0], want[name][0], *TOL["loc"])
        rep.close("scale/" + name, scale[0], want[name][1], *TOL["loc"])
    rep.finish()
    mean = daa.mean_reconstructions(model, data, 16)
    for name in data:       # (stochastic: shapes and finiteness; the per-sample scale is averaged)
        loc, scale = mean[name]
        assert loc.shape == scale.shape == (n, data[name].shape[1])
        assert bool(torch.isfinite(loc).all()) and bool((scale > 0).all())
    sv = torch.randn(3, n, 7, generator=g).cuda()
    got = daa.perturbed_reconstructions(model, data, sv, "likelihood", sample_latents=False)
    cdata = data["clinical"].clone()
    cdata[:, 2] = sv[1, :, 2]
    ref = model({"clinical": cdata, "rois": data["rois"]}, sample_latents=False)["rec"]["rois"].loc
    assert torch.equal(got[:, 2, 1], ref)
