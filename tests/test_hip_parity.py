"""GPU: the HIP path (through the C ABI) against the CPU oracle on the same
seeded inputs and injected eps, for every golden case -- intermediates, loss
terms, gradients, parameters and Adam state over several steps."""
from collections import OrderedDict

import pytest
import torch

import mopoe_oracle as mo
from golden_util import Fixture, case_names
from hip_util import Report, TOL, compare_forward, make_engine

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["four_rows", "sixteen_rows"])
def row_groups_form(request, monkeypatch):
    """Every case under both forms of the fused launch: the four-row groups small batches
    of <= 2 modalities get by default, and (MOPOE_QUAD=0) the 16-row groups."""
    if request.param == "sixteen_rows":
        monkeypatch.setenv("MOPOE_QUAD", "0")
    else:
        monkeypatch.delenv("MOPOE_QUAD", raising=False)


@pytest.mark.parametrize("case", case_names())
def test_train_steps_match_oracle(case):
    fx = Fixture(case)
    cfg = fx.cfg
    spec, eng = make_engine(cfg)
    params = mo.init_params(cfg, 0)
    state = mo.adam_init(params)
    rep = Report(case)
    adam_steps = OrderedDict((k, 0) for k in params)   # torch: one count per parameter
    for step in range(fx.steps):
        x = fx.inputs_at(step)      # (mixed-mask cases: the batch's modalities change)
        noise = fx.noise(step)
        # the oracle steps from the HIP path's current parameters, so every
        # step is compared on identical weights
        params = OrderedDict((k, v.cpu().clone()) for k, v in eng.named_params().items())
        m_before = OrderedDict((k, v.cpu().clone()) for k, v in
                               spec.param_views(eng.exp_avg).items())
        v_before = OrderedDict((k, v.cpu().clone()) for k, v in
                               spec.param_views(eng.exp_avg_sq).items())
        state = {"step": adam_steps, "exp_avg": m_before, "exp_avg_sq": v_before}
        out, grads = mo.train_step(params, cfg, x, noise, state)
        # (general topologies: the dropout keep masks the oracle drew are injected too)
        plan, ws = eng.train_step(x, eps=noise.tape, masks=noise.mask_tape or None)
        torch.cuda.synchronize()
        p = "step%d/" % step
        compare_forward(rep, spec, eng, plan, ws, out, prefix=p, check_scale=False)
        for k, g in grads.items():
            rep.close_scaled(p + "grad/" + k, eng.grad_views[k], g, TOL["grad"])
        # parameters after the fused Adam update vs torch-semantics Adam on
        # the ORACLE's gradient; an element whose gradient is ~0 may move by
        # +-lr in either direction, so compare through the moments instead
        for k in grads:
            rep.close_scaled(p + "exp_avg/" + k, spec.param_views(eng.exp_avg)[k],
                             state["exp_avg"][k], TOL["grad"])
            rep.close_scaled(p + "exp_avg_sq/" + k,
                             spec.param_views(eng.exp_avg_sq)[k],
                             state["exp_avg_sq"][k], TOL["moment2"])
        # parameters: where the gradient is not ~0 the update is well
        # conditioned (|dp| <= lr); elsewhere sign(g) decides and 1e-8-level
        # gradient noise may flip it
        new = eng.named_params()
        for k, g in grads.items():
            mask = g.abs() > 1e-6
            rep.close(p + "param/" + k, new[k].cpu()[mask], params[k][mask],
                      *TOL["param1"])
            # the Normal's scale now reflects the updated logvar
        res = eng.results(plan, ws)
        for k in res["rec"]:
            if cfg.sample_scale:    # the head's (N, d) scale of THIS forward
                rep.close(p + "rec/%s/scale" % k, res["rec"][k].scale,
                          out["results"]["rec"][k][1], 2e-5, 1e-6)
            else:
                rep.close(p + "rec/%s/scale" % k, res["rec"][k].scale[0],
                          (params["decoders.%s.logvar" % k][0] * 0.5).exp(), 1e-6, 1e-7)
        assert eng.step_count() == step + 1
        # the device's per-modality Adam counts are torch's per-parameter ones
        for name, t in eng.adam_steps().items():
            assert t == adam_steps["encoders.%s.class_mu.weight" % name], (name, t)
        eng.check_valid(sync=True)
    rep.finish()


@pytest.mark.parametrize("case", case_names(fwd=True))
def test_forward_flags_match_oracle(case):
    fx = Fixture(case)
    spec, eng = make_engine(fx.cfg)
    params = mo.init_params(fx.cfg, 0)
    noise = fx.noise(0)
    x = fx.inputs()
    with torch.no_grad():
        out = mo.forward(params, fx.cfg, x, noise,
                         sample_latents=fx.meta["sample_latents"],
                         use_expert=fx.meta["use_expert"])
    plan, ws = eng.forward(x, sample=fx.meta["sample_latents"],
                           use_expert=fx.meta["use_expert"], eps=noise.tape)
    torch.cuda.synchronize()
    rep = Report(case)
    compare_forward(rep, spec, eng, plan, ws, out)
    rep.finish()
