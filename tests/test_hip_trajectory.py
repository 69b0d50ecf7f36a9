"""GPU: ELBO match over a free-running trajectory at BASELINE size (SURVEY.md 8d:
"after 1 and 20 steps").  Unlike tests/test_hip_parity.py, the oracle is NOT
re-synchronised to the HIP path's parameters between steps: both start from the
same weights and see the same batches and the same injected eps, and run twenty
Adam steps on their own.

Stated tolerances (the per-step float32 bounds of hip_util.TOL, kept over the
whole trajectory): every step's total loss, joint divergence and per-modality
NLL within 2e-5 relative; after step 1 and step 20 the joint and subset
posteriors (mu, logvar) within 2e-5 (rel + abs) and the reconstructions within
2e-5 rel + 5e-5 abs.  Measured on MI355X: <= 2.2e-6 absolute on the posteriors
and <= 1e-7 relative on the loss after twenty steps, in all three configurations."""
from collections import OrderedDict

import pytest
import torch

import mopoe_oracle as mo
from hip_util import Report, make_engine

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["four_rows", "sixteen_rows"])
def row_groups_form(request, monkeypatch):
    """Every case under both forms of the fused launch: the four-row groups small batches
    of <= 2 modalities get by default, and (MOPOE_QUAD=0) the 16-row groups."""
    if request.param == "sixteen_rows":
        monkeypatch.setenv("MOPOE_QUAD", "0")
    else:
        monkeypatch.delenv("MOPOE_QUAD", raising=False)

CASES = {
    # configs[1] / C1: 2-modality joint_elbo, dims 7+444, latent 20, batch 256
    "C1_joint_elbo_bs256": dict(names=["clinical", "rois"], dims=[7, 444], style=[3, 20],
                                method="joint_elbo", n=256),
    # configs[2] / C3: method poe, batch 1024
    "C3_poe_bs1024": dict(names=["clinical", "rois"], dims=[7, 444], style=[3, 20],
                          method="poe", n=1024),
    # configs[4] / C5: four modalities, 15 subsets, batch 512
    "C5_four_mods_bs512": dict(names=["clinical", "rois", "m3", "m4"], dims=[7, 444, 128, 64],
                               style=[3, 3, 3, 3], method="joint_elbo", n=512),
}
STEPS = 20


@pytest.mark.parametrize("case", sorted(CASES))
def test_twenty_free_running_steps_match_oracle(case):
    c = CASES[case]
    cfg = mo.Config(c["names"], c["dims"], c["style"], method=c["method"])
    spec, eng = make_engine(cfg)
    params = mo.init_params(cfg, 0)
    state = mo.adam_init(params)
    rng = mo.noise_rng(31)
    rep = Report(case)
    for step in range(STEPS):
        x = mo.make_inputs(cfg.names, cfg.input_dim, c["n"], seed=1000 + step)
        noise = mo.Noise(generator=rng)
        out, _ = mo.train_step(params, cfg, x, noise, state)   # records its eps on the tape
        plan, ws = eng.train_step(x, eps=noise.tape)
        torch.cuda.synchronize()
        sc = eng.scalars(plan, ws)
        res = eng.results(plan, ws)
        p = "step%02d/" % (step + 1)
        rep.close(p + "total_loss", sc["total_loss"], out["total_loss"], 2e-5, 0)
        rep.close(p + "joint_divergence", res["joint_divergence"],
                  out["results"]["joint_divergence"], 2e-5, 1e-5)
        for k, v in out["log_probs"].items():
            rep.close(p + "log_probs/" + k, sc["log_probs"][k], v, 2e-5, 0)
        if step in (0, STEPS - 1):
            lat, lat_o = res["latents"], out["results"]["latents"]
            rep.close(p + "joint/mu", lat["joint"][0], lat_o["joint"][0], 2e-5, 2e-5)
            rep.close(p + "joint/logvar", lat["joint"][1], lat_o["joint"][1], 2e-5, 2e-5)
            for k, (mu, lv) in lat_o["subsets"].items():
                rep.close(p + "subsets/%s/mu" % k, lat["subsets"][k][0], mu, 2e-5, 2e-5)
                rep.close(p + "subsets/%s/logvar" % k, lat["subsets"][k][1], lv, 2e-5, 2e-5)
            for k, (loc, scale) in out["results"]["rec"].items():
                rep.close(p + "rec/%s/loc" % k, res["rec"][k].loc, loc, 2e-5, 5e-5)
    print("\n".join("%-40s max|err| %.3e  (%.2f of tol)" % r for r in rep.rows
                    if r[0].startswith(("step01/", "step20/"))))
    rep.finish()
