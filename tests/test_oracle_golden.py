"""CPU: the oracle (oracle/mopoe_oracle.py) against the golden fixtures that
oracle/make_golden.py recorded from the reference itself.  This is what pins
the oracle (the reference has no tests of its own)."""
import json
from collections import OrderedDict

import numpy as np
import pytest
import torch

import mopoe_oracle as mo
from golden_util import (Fixture, assert_close, case_names, check_digests,
                         GOLDEN_DIR)

# oracle and reference run the same torch CPU ops in the same order
RTOL, ATOL = 1e-6, 1e-6


def compare_results(fx, out, prefix="step0"):
    res = out["results"] if "results" in out else out
    lat = res["latents"]
    if "total_loss" in out:
        assert_close(out["total_loss"], fx.get(prefix + "/total_loss"),
                     RTOL, ATOL, "total_loss")
        for k, v in out["log_probs"].items():
            assert_close(v, fx.get(prefix + "/log_probs/" + k), RTOL, ATOL, k)
        assert ["step0/klds/" + k for k in out["klds"]] == \
            fx.keys(prefix + "/klds/")
        for k, v in out["klds"].items():
            assert_close(v, fx.get(prefix + "/klds/" + k), RTOL, ATOL, k)
    assert_close(res["joint_divergence"], fx.get(prefix + "/joint_divergence"),
                 RTOL, ATOL, "joint_divergence")
    assert_close(res["individual_divs"], fx.get(prefix + "/individual_divs"),
                 RTOL, ATOL, "individual_divs")
    assert_close(lat["weights"], fx.get(prefix + "/weights"), RTOL, ATOL, "w")
    if not fx.full:
        return
    for k, (mu, lv) in lat["modalities"].items():
        if mu is None:
            assert not fx.has(prefix + "/modalities/" + k + "/mu")
            continue
        assert_close(mu, fx.get(prefix + "/modalities/" + k + "/mu"),
                     RTOL, ATOL, k)
        assert_close(lv, fx.get(prefix + "/modalities/" + k + "/logvar"),
                     RTOL, ATOL, k)
    want_subsets = [k.split("/")[2] for k in fx.keys(prefix + "/subsets/")
                    if k.endswith("/mu")]
    assert list(lat["subsets"].keys()) == want_subsets
    for k, (mu, lv) in lat["subsets"].items():
        assert_close(mu, fx.get(prefix + "/subsets/" + k + "/mu"),
                     RTOL, ATOL, k)
        assert_close(lv, fx.get(prefix + "/subsets/" + k + "/logvar"),
                     RTOL, ATOL, k)
    if fx.has(prefix + "/mus"):
        assert_close(lat["mus"], fx.get(prefix + "/mus"), RTOL, ATOL, "mus")
        assert_close(lat["logvars"], fx.get(prefix + "/logvars"), RTOL, ATOL,
                     "logvars")
    assert_close(lat["joint"][0], fx.get(prefix + "/joint/mu"), RTOL, ATOL,
                 "joint mu")
    assert_close(lat["joint"][1], fx.get(prefix + "/joint/logvar"), RTOL,
                 ATOL, "joint logvar")
    for k, (loc, scale) in res["rec"].items():
        assert_close(loc, fx.get(prefix + "/rec/" + k + "/loc"), RTOL, ATOL,
                     "loc " + k)
        # torch.distributions.Normal broadcasts the (1, d) scale to (N, d)
        assert_close(scale.expand_as(loc),
                     fx.get(prefix + "/rec/" + k + "/scale"), RTOL, ATOL,
                     "scale " + k)


@pytest.mark.parametrize("case", case_names())
def test_train_steps_match_reference(case):
    fx = Fixture(case)
    cfg = fx.cfg
    params = mo.init_params(cfg, 0)
    state = mo.adam_init(params)
    for step in range(fx.steps):
        x = fx.inputs_at(step)
        noise = fx.noise(step)
        out, grads = mo.train_step(params, cfg, x, noise, state)
        fx.check_noise(step, noise)
        if fx.every_step:   # which parameters torch left without a gradient
            none = json.loads(str(fx.z["step%d/grad_none" % step]))
            assert sorted(none) == sorted(set(params) - set(grads))
        if step == 0:
            compare_results(fx, out)
            none = json.loads(str(fx.z["step0/grad_none"]))
            assert sorted(none) == sorted(set(params) - set(grads))
            check_digests(fx, "step0/grads", grads, 1e-4, 1e-6)
        else:
            assert_close(out["total_loss"],
                         fx.get("step%d/total_loss" % step), 1e-5, 1e-6)
        if step in (0, fx.steps - 1) or fx.every_step:
            check_digests(fx, "after%d/params" % (step + 1), params,
                          1e-6 if step == 0 else 1e-4, 1e-6)
    if fx.every_step:       # torch's per-parameter Adam step counts
        want = json.loads(str(fx.z["final/adam_steps"]))
        assert {k: v for k, v in state["step"].items() if v} == want
    touched = OrderedDict((k, v) for k, v in state["exp_avg"].items()
                          if fx.has("final/exp_avg/" + k + "/stats"))
    # several steps of Adam amplify 1-ulp differences in early gradients
    check_digests(fx, "final/exp_avg", touched, 2e-4, 1e-7)
    touched = OrderedDict((k, v) for k, v in state["exp_avg_sq"].items()
                          if fx.has("final/exp_avg_sq/" + k + "/stats"))
    check_digests(fx, "final/exp_avg_sq", touched, 2e-4, 1e-10)


@pytest.mark.parametrize("case", case_names(fwd=True))
def test_forward_flags_match_reference(case):
    fx = Fixture(case)
    params = mo.init_params(fx.cfg, 0)
    noise = fx.noise(0)
    with torch.no_grad():
        res = mo.forward(params, fx.cfg, fx.inputs(), noise,
                         sample_latents=fx.meta["sample_latents"],
                         use_expert=fx.meta["use_expert"])
    fx.check_noise(0, noise)
    compare_results(fx, res)


def test_l0_functions_match_reference():
    z = np.load(GOLDEN_DIR + "/l0_functions.npz")
    t = lambda k: torch.from_numpy(z[k])
    for E in (1, 2, 3, 5):
        pm, plv = mo.poe(t("poe/%d/mu" % E), t("poe/%d/logvar" % E))
        assert_close(pm, t("poe/%d/out_mu" % E), RTOL, ATOL)
        assert_close(plv, t("poe/%d/out_logvar" % E), RTOL, ATOL)
    assert_close(mo.calc_kl_divergence(t("kl/mu"), t("kl/logvar")),
                 t("kl/out"), RTOL, ATOL)
    assert_close(mo.calc_kl_divergence(t("kl/mu"), t("kl/logvar"), 33),
                 t("kl/out_norm"), RTOL, ATOL)
    for K, N in ((3, 256), (15, 512), (3, 37), (7, 5), (2, 1), (1, 8)):
        p = "mix/%d_%d/" % (K, N)
        w = mo.reweight_weights((1 / float(K)) * torch.ones(K))
        m_sel, l_sel = mo.mixture_component_selection(t(p + "mus"),
                                                      t(p + "logvars"), w)
        assert torch.equal(m_sel, t(p + "out_mu"))
        assert torch.equal(l_sel, t(p + "out_logvar"))
        gd, klds = mo.calc_group_divergence_moe(t(p + "mus"), t(p + "logvars"),
                                                w, N)
        assert_close(gd, t(p + "group_div"), RTOL, ATOL)
        assert_close(klds, t(p + "klds"), RTOL, ATOL)


def test_mixture_bounds_known_answers():
    # SURVEY.md section 8a row a5
    w3 = mo.reweight_weights(torch.ones(3) / 3.0)
    assert mo.mixture_bounds(256, w3) == ([0, 85, 170], [85, 170, 256])
    w15 = mo.reweight_weights(torch.ones(15) / 15.0)
    s, e = mo.mixture_bounds(512, w15)
    assert s[1] - s[0] == 34 and e[-1] - s[-1] == 36


def test_subset_order():
    assert list(mo.set_subsets(["clinical", "rois"]).keys()) == \
        ["", "clinical", "rois", "clinical_rois"]
    assert list(mo.set_subsets(["zeta", "alpha"]).keys()) == \
        ["", "zeta", "alpha", "alpha_zeta"]
    assert len(mo.set_subsets(["a", "b", "c", "d"])) == 16
