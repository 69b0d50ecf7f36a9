"""GPU, SURVEY.md section 8f row f3: the per-step scalar log.  The kernels write ONE
stats vector per step -- also straight into pinned host memory (`stats_host`) -- that
holds everything the reference's TBLogger writes per step (utils/TBLogger.py:84-96:
loss, log-probabilities, KL of every subset, group divergence, mean mu / mean logvar of
every latent).  Checked entry by entry against the oracle, for training steps and for
the evaluation routine (run_epochs.test: forward only, latents still sampled)."""
from collections import OrderedDict

import pytest
import torch

import mopoe_amd as mm
import mopoe_oracle as mo
from hip_util import Report, TOL, make_engine
from surface_util import make_experiment, run_epochs

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["four_rows", "sixteen_rows"])
def row_groups_form(request, monkeypatch):
    """Every case under both forms of the fused launch: the four-row groups small batches
    of <= 2 modalities get by default, and (MOPOE_QUAD=0) the 16-row groups."""
    if request.param == "sixteen_rows":
        monkeypatch.setenv("MOPOE_QUAD", "0")
    else:
        monkeypatch.delenv("MOPOE_QUAD", raising=False)
L = mm._lib

CASES = {
    "c1_joint": dict(names=["clinical", "rois"], dims=[7, 444], style=[3, 20],
                     method="joint_elbo", present=None, n=256),
    "c1_only_rois": dict(names=["clinical", "rois"], dims=[7, 444], style=[3, 20],
                         method="joint_elbo", present=["rois"], n=37),
    "c3_poe": dict(names=["clinical", "rois"], dims=[7, 444], style=[3, 20],
                   method="poe", present=None, n=64),
    "c5_missing": dict(names=["a", "b", "c", "d"], dims=[7, 444, 128, 64], style=[3, 3, 3, 3],
                       method="joint_elbo", present=["a", "c", "d"], n=40),
    "nofact_moe": dict(names=["clinical", "rois"], dims=[7, 444], style=[3, 20],
                       method="moe", present=None, n=48, factorized=False),
}


def oracle_log(cfg, out):
    """The scalar set of TBLogger.add_basic_logs from the oracle's step output."""
    lat = out["results"]["latents"]["modalities"]
    return {"Loss": {"loss": out["total_loss"]},
            "LogProb": out["log_probs"], "KLD": out["klds"],
            "group_divergence": {"group_div": out["results"]["joint_divergence"]},
            "mu": OrderedDict((k, v[0].mean()) for k, v in lat.items() if v[0] is not None),
            "logvar": OrderedDict((k, v[1].mean()) for k, v in lat.items() if v[1] is not None)}


def compare_logs(rep, got, want, prefix=""):
    assert list(got) == ["Loss", "LogProb", "KLD", "group_divergence", "mu", "logvar"]
    for tag, vals in want.items():
        assert list(got[tag].keys()) == list(vals.keys()), (tag, list(got[tag]), list(vals))
        for k, v in vals.items():
            if tag in ("mu", "logvar"):
                # a mean over N * D values of either sign: float32 bound of the SUM
                rep.close(prefix + tag + "/" + k, got[tag][k], float(v), 2e-5, 2e-6)
            else:
                rep.close(prefix + tag + "/" + k, got[tag][k], float(v), *TOL["scalar"])


@pytest.mark.parametrize("case", sorted(CASES))
def test_every_logged_scalar_of_a_training_step(case):
    c = CASES[case]
    cfg = mo.Config(c["names"], c["dims"], c["style"], method=c["method"],
                    factorized=c.get("factorized", True))
    spec, eng = make_engine(cfg)
    params = mo.init_params(cfg, 0)
    x = mo.make_inputs(cfg.names, cfg.input_dim, c["n"], seed=3, present=c["present"])
    noise = mo.Noise(generator=mo.noise_rng(8))
    out, _ = mo.loss_and_grads(params, cfg, x, noise)
    host = torch.full((L.NUM_STATS,), float("nan")).pin_memory()
    plan, ws = eng.train_step(x, eps=noise.tape, stats_host=host)
    torch.cuda.synchronize()
    assert torch.equal(host, ws.stats.cpu())          # the pinned copy is the same vector
    rep = Report(case)
    compare_logs(rep, eng.log_scalars(plan, host), oracle_log(cfg, out))
    # style KLs (logged by the loop through klds_style) and the unused entries
    sc = eng.scalars(plan, ws)
    for k, v in out["klds_style"].items():
        rep.close("klds_style/" + k, sc["klds_style"][k], v, *TOL["scalar"])
    used = {L.STAT_TOTAL_LOSS, L.STAT_JOINT_DIV}
    used |= {L.STAT_KLD_SUBSET + s for s in plan.avail_idx}
    used |= {L.STAT_KLD_STYLE + m for m in plan.present_idx if spec.has_style(m)}
    used |= {L.STAT_NLL + j for j in range(len(plan.jobs))}
    for m in plan.present_idx:
        used |= {L.STAT_LATENT_MEAN + 4 * m + k for k in ((0, 1, 2, 3) if spec.has_style(m)
                                                          else (2, 3))}
    rest = [i for i in range(L.NUM_STATS) if i not in used]
    assert float(host[rest].abs().max()) == 0.0
    rep.finish()


def test_evaluation_routine_matches_oracle():
    """run_epochs.test's basic_routine_epoch under no_grad (reference run_epochs.py:
    187-209): forward + loss terms, latents sampled, no parameter touched."""
    cfg = mo.Config(["clinical", "rois"], [7, 444], [3, 20], method="poe")
    exp = make_experiment(cfg, "cuda")
    model = exp.models
    params = mo.init_params(cfg, 0)
    model.load_state_dict(params)
    model.eval()
    x = mo.make_inputs(cfg.names, cfg.input_dim, 50, seed=21)
    noise = mo.Noise(generator=mo.noise_rng(22))
    noise.dtype = torch.float32
    with torch.no_grad():
        out = mo.basic_routine_epoch(params, cfg, x, noise)
    eng = model.engine
    before = eng.params.clone()
    orig = eng.forward
    eng.forward = lambda b, **kw: orig(b, eps=noise.tape, **kw)
    with torch.no_grad():
        res = run_epochs.basic_routine_epoch(
            exp, 0, (OrderedDict((k, v.double()) for k, v in x.items()), None, {}))
    eng.forward = orig
    torch.cuda.synchronize()
    rep = Report("eval")
    rep.close("total_loss", res["total_loss"], out["total_loss"], *TOL["scalar"])
    assert not res["total_loss"].requires_grad
    for k, v in out["log_probs"].items():
        rep.close("log_probs/" + k, res["log_probs"][k], v, *TOL["scalar"])
    for k, v in out["klds"].items():
        rep.close("klds/" + k, res["klds"][k], v, *TOL["scalar"])
    rep.close("joint_divergence", res["results"]["joint_divergence"],
              out["results"]["joint_divergence"], *TOL["scalar"])
    assert torch.equal(before, eng.params) and eng.step_count() == 0
    rep.finish()
