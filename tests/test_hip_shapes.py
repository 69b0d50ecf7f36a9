"""GPU: shapes and sizes the golden cases do not reach, HIP path (through the C ABI)
against the CPU oracle on the same seeded inputs and eps -- five modalities (31
subsets), inputs wider than one K chunk of the encoder layer, odd latent sizes, tiny
and ragged batches, a frozen output scale.  The oracle is pinned to the reference by
tests/golden (tests/test_oracle_golden.py); these cases are pinned to the oracle only."""
from collections import OrderedDict

import numpy as np
import pytest
import torch

import mopoe_oracle as mo
from hip_util import Report, TOL, compare_forward, make_engine

pytestmark = pytest.mark.gpu

CASES = {
    # name: (names, dims, style, kwargs, N, present)
    "five_modalities": (list("abcde"), [7, 444, 128, 64, 33], [3, 20, 8, 8, 4], {}, 40, None),
    "five_modalities_poe": (list("abcde"), [7, 96, 128, 64, 33], [3, 6, 8, 8, 4],
                            dict(method="poe"), 24, None),
    "five_modalities_three_present": (list("abcde"), [7, 444, 128, 64, 33], [3, 20, 8, 8, 4],
                                      {}, 19, ["a", "c", "e"]),
    "wide_inputs": (["p", "q"], [700, 1100], [5, 12], dict(class_dim=32), 48, None),
    "odd_latents": (["p", "q"], [7, 444], [5, 1], dict(class_dim=7), 33, None),
    "one_row": (["clinical", "rois"], [7, 444], [3, 20], {}, 1, None),
    "fifteen_rows_moe": (["clinical", "rois"], [7, 444], [3, 20], dict(method="moe"), 15, None),
    "frozen_output_scale": (["clinical", "rois"], [7, 444], [3, 20],
                            dict(learn_output_scale=False), 32, None),
    "not_factorized_poe": (["clinical", "rois"], [7, 444], [3, 20],
                           dict(method="poe", factorized=False), 50, None),
}


def _noise(cfg, plan, n, seed):
    g = np.random.Generator(np.random.PCG64(seed))
    tape = []
    for kind, j in plan.noise_slots:
        m = plan.jobs[j][0]
        w = cfg.class_dim if kind == "content" else cfg.style_dim[m]
        tape.append(torch.from_numpy(g.standard_normal((n, w)).astype(np.float32)))
    return tape


@pytest.mark.parametrize("rows", [1, 2, 4, 8])
@pytest.mark.parametrize("case", ["odd_latents", "five_modalities_three_present",
                                  "not_factorized_poe", "fifteen_rows_moe"])
def test_small_row_groups_match_oracle(case, rows, monkeypatch):
    """The fallback for models whose 16-row tiles do not fit the LDS (groups of 8, 4, 2
    or 1 rows), forced here on models that would not need it."""
    monkeypatch.setenv("MOPOE_ROWS_PER_GROUP", str(rows))
    test_train_step_matches_oracle(case)


@pytest.mark.parametrize("case", sorted(CASES))
def test_train_step_matches_oracle(case):
    names, dims, style, kw, n, present = CASES[case]
    cfg = mo.Config(names, dims, style, **kw)
    spec, eng = make_engine(cfg)
    x = mo.make_inputs(names, dims, n, seed=11, present=present)
    rep = Report(case)
    for step in range(2):
        params = OrderedDict((k, v.cpu().clone()) for k, v in eng.named_params().items())
        state = {"step": step,
                 "exp_avg": OrderedDict((k, v.cpu().clone()) for k, v in
                                        spec.param_views(eng.exp_avg).items()),
                 "exp_avg_sq": OrderedDict((k, v.cpu().clone()) for k, v in
                                           spec.param_views(eng.exp_avg_sq).items())}
        plan = spec.plan(list(x.keys()), n, True, None, True, True)
        tape = _noise(cfg, plan, n, 100 + step)
        before = OrderedDict((k, v.clone()) for k, v in params.items())
        out, grads = mo.train_step(params, cfg, x, mo.Noise(tape=tape), state)
        plan, ws = eng.train_step(x, eps=tape)
        torch.cuda.synchronize()
        p = "step%d/" % step
        compare_forward(rep, spec, eng, plan, ws, out, prefix=p, check_scale=False)
        for k, g in grads.items():
            rep.close_scaled(p + "grad/" + k, eng.grad_views[k], g, TOL["grad"])
            rep.close_scaled(p + "exp_avg/" + k, spec.param_views(eng.exp_avg)[k],
                             state["exp_avg"][k], TOL["grad"])
            rep.close_scaled(p + "exp_avg_sq/" + k, spec.param_views(eng.exp_avg_sq)[k],
                             state["exp_avg_sq"][k], TOL["moment2"])
        new = eng.named_params()
        for k, g in grads.items():
            mask = g.abs() > 1e-6
            rep.close(p + "param/" + k, new[k].cpu()[mask], params[k][mask], *TOL["param1"])
        for k in params:       # parameters without a gradient must not move
            if k not in grads:
                assert torch.equal(new[k].cpu(), before[k]), k
    rep.finish()
