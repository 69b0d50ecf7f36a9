"""GPU: the weight gradients of a LARGE batch (>= 4,096 rows: 64 x 64 tiles over slices of
the batch rows + an ordered second launch, csrc/mopoe_wgrad_big.inc) against the oracle and
against the one-launch form -- ragged row counts, gathered inputs, method poe's doubled
decoder rows, four modalities, a frozen decoder scale."""
from collections import OrderedDict

import pytest
import torch

import mopoe_amd as mm
import mopoe_oracle as mo
from hip_util import Report, TOL, compare_forward, make_engine

pytestmark = pytest.mark.gpu
L = mm._lib
C1 = dict(names=["clinical", "rois"], input_dim=[7, 444], style_dim=[3, 20])
C5 = dict(names=["clinical", "rois", "snps", "tracts"], input_dim=[7, 444, 128, 64], style_dim=[3, 3, 3, 3])


@pytest.mark.parametrize("base,method,n,extra", [
    (C1, "joint_elbo", 4096 + 37, {}), (C1, "poe", 4100, {}), (C5, "joint_elbo", 4096, {}),
    (C1, "moe", 5000, dict(learn_output_scale=False, factorized=False)),
    # the opt-in bfloat16 operands of the encoder layer (mopoe_step.gemm_operands) against
    # their definition (oracle: roundings of x and W, float32 sums, float32 backward) -- the
    # products of bfloat16 values are exact in float32, so the tolerances stay what they are
    (C1, "joint_elbo", 4096 + 37, dict(gemm_operands="bf16"))],
    ids=["c1_ragged", "c1_poe", "c5", "c1_moe_frozen_scale", "c1_bf16_operands"])
@pytest.mark.parametrize("form", ["split", "one_launch"])
def test_large_batch_step_matches_oracle(base, method, n, extra, form):
    cfg = mo.Config(method=method, **base, **extra)
    spec, eng = make_engine(cfg)
    if form == "one_launch":        # (k_wgrad at the same size: the form the split one replaces)
        slots = 2 if method == "poe" else 1
        eng.workspace(n, slots, True).wgrad_scratch = False
    params = mo.init_params(cfg, 0)
    state = mo.adam_init(params)
    x = mo.make_inputs(cfg.names, cfg.input_dim, n, seed=3)
    noise = mo.Noise(generator=mo.noise_rng(4))
    out, grads = mo.train_step(params, cfg, x, noise, state)
    plan, ws = eng.train_step(x, eps=noise.tape)
    torch.cuda.synchronize()
    eng.check_valid(sync=True)
    if form == "split":
        assert torch.is_tensor(ws.wgrad_scratch) and ws.wgrad_scratch.numel() == \
            L.lib.mopoe_wgrad_scratch_floats(spec.c_model, plan.c_step) > 0  # (the form under test ran)
    else:
        assert ws.wgrad_scratch is False
    rep = Report("large batch %s n=%d" % (method, n))
    compare_forward(rep, spec, eng, plan, ws, out, check_scale=False)
    # With millions of pre-activations per step some land within rounding of ReLU's kink, and
    # two float32 summation orders then disagree on [pre > 0] for that (row, unit): the whole
    # gradient row of that hidden unit (first-layer weights + bias; nothing else depends on
    # the first layer's mask) moves by a finite amount.  Such units -- |pre| < 3e-6 anywhere
    # in the batch, from the oracle's float64 pre-activations -- are left out of the comparison.
    init = mo.init_params(cfg, 0)
    near_kink = {}
    for name in x:
        e = "encoders.%s.shared_encoder.0." % name
        xs, ws_ = x[name], init[e + "weight"]
        if cfg.gemm_operands == "bf16":
            xs, ws_ = xs.bfloat16(), ws_.bfloat16()
        pre = xs.double() @ ws_.double().t() + init[e + "bias"].double()
        near_kink[name] = (pre.abs() < 3e-6).any(0)
    skipped = sum(int(v.sum()) for v in near_kink.values())
    assert skipped <= 32, skipped
    for k, g in grads.items():
        keep = slice(None)
        if ".shared_encoder.0." in k:
            keep = ~near_kink[k.split(".")[1]]
        rep.close_scaled("grad/" + k, eng.grad_views[k].cpu()[keep], g[keep], TOL["grad"])
        rep.close_scaled("exp_avg/" + k, spec.param_views(eng.exp_avg)[k].cpu()[keep],
                         state["exp_avg"][k][keep], TOL["grad"])
        rep.close_scaled("exp_avg_sq/" + k, spec.param_views(eng.exp_avg_sq)[k].cpu()[keep],
                         state["exp_avg_sq"][k][keep], TOL["moment2"])
    for k in set(params) - set(grads):           # a frozen parameter stays put
        assert torch.equal(eng.named_params()[k].cpu(), mo.init_params(cfg, 0)[k]), k
    rep.finish()


def test_split_form_agrees_with_the_one_launch_form_and_keeps_the_weight_copies():
    """Same gradients up to the summation order (gathered rows included), the fragment-major
    weight copies follow the update, and a following small batch runs the four-row form on
    them."""
    cfg = mo.Config(**C1)
    _, a = make_engine(cfg)
    _, b = make_engine(cfg)
    n = 8192
    pool = mo.make_inputs(cfg.names, cfg.input_dim, 9000, seed=5)
    idx = torch.randperm(9000, generator=torch.Generator().manual_seed(1))[:n].to(torch.int32)
    eps = mo.Noise(generator=mo.noise_rng(6))
    mo.forward(mo.init_params(cfg, 0), cfg, OrderedDict((k, v[idx.long()]) for k, v in pool.items()), eps)
    xa = {k: v.cuda() for k, v in pool.items()}
    _, wa = a.train_step(xa, eps=eps.tape, row_index=idx.cuda())
    torch.cuda.synchronize()
    wb = b.workspace(n, 1, True)
    wb.wgrad_scratch = False                      # (the one-launch form)
    b.train_step(xa, eps=eps.tape, row_index=idx.cuda())
    torch.cuda.synchronize()
    assert torch.is_tensor(wa.wgrad_scratch) and wb.wgrad_scratch is False
    scale = b.grads.abs().max().item()
    assert (a.grads - b.grads).abs().max().item() < 2e-6 * scale
    # (the same forward; the split form adds the row groups' partial sums in 64 slices
    #  first, the one-launch form walks them in one chain: float32 rounding of long sums)
    torch.testing.assert_close(wa.stats, wb.stats, rtol=5e-6, atol=1e-6)
    ref = a.wfrag.clone()
    a.refresh_wfrag()
    torch.cuda.synchronize()
    assert torch.equal(ref, a.wfrag)
    small = mo.make_inputs(cfg.names, cfg.input_dim, 256, seed=7)
    e2 = mo.Noise(generator=mo.noise_rng(8))
    params = OrderedDict((k, v.cpu().clone()) for k, v in a.named_params().items())
    out, grads = mo.loss_and_grads(params, cfg, small, e2)
    a.train_step(small, eps=e2.tape, apply_adam=False)
    torch.cuda.synchronize()
    rep = Report("four-row form after a large batch")
    for k, g in grads.items():
        rep.close_scaled("grad/" + k, a.grad_views[k], g, TOL["grad"])
    rep.finish()


def _near_kink(cfg, params, x, eps=3e-6):
    """Hidden units whose pre-activation is within `eps` of ReLU's kink anywhere in the batch
    (two float32 summation orders may disagree on their mask: see the first test)."""
    out = {}
    for name in x:
        e = "encoders.%s.shared_encoder.0." % name
        pre = x[name].double() @ params[e + "weight"].double().t() + params[e + "bias"].double()
        out[name] = (pre.abs() < eps).any(0)
    return out


def test_scratch_follows_the_batch_modalities():
    """One workspace serves every batch of its shape, but the scratch of the split weight-
    gradient launches depends on WHICH modalities the batch holds (4,096 rows: clinical only
    719,360 floats, rois only 3,013,120, both 3,668,480): a cohort epoch of missing-modality
    batches used to size it from the first batch and let later ones write past it.  The
    engine now asks per plan and grows the tensor; every step's gradients against the oracle
    (same parameters every step: the steps are not applied)."""
    cfg = mo.Config(**C1)
    spec, eng = make_engine(cfg)
    n = 4096
    params = mo.init_params(cfg, 0)
    full = mo.make_inputs(cfg.names, cfg.input_dim, n, seed=21)
    seen = []
    for step, names in enumerate((["clinical"], ["rois"], ["clinical", "rois"], ["clinical"])):
        x = OrderedDict((k, full[k]) for k in names)
        noise = mo.Noise(generator=mo.noise_rng(30 + step))
        out, grads = mo.loss_and_grads(params, cfg, x, noise)
        guard = torch.full((1 << 20,), 7.0, device="cuda")     # (allocated right behind: a write
        plan, ws = eng.train_step(x, eps=noise.tape, apply_adam=False)   # past the scratch lands here)
        torch.cuda.synchronize()
        eng.check_valid(sync=True)
        need = L.lib.mopoe_wgrad_scratch_floats(spec.c_model, plan.c_step)
        assert need > 0 and ws.wgrad_scratch.numel() >= need
        assert bool((guard == 7.0).all())
        seen.append(need)
        kink = _near_kink(cfg, params, x)
        assert sum(int(v.sum()) for v in kink.values()) <= 32
        rep = Report("scratch per plan: %s" % "+".join(names))
        compare_forward(rep, spec, eng, plan, ws, out, check_scale=False)
        for k, g in grads.items():
            m = k.split(".")[1]
            got = eng.grad_views[k].cpu()
            if ".shared_encoder.0." in k:           # rows of W1 / b1: the unit itself
                got, g = got[~kink[m]], g[~kink[m]]
            elif k.startswith("encoders.") and k.endswith(".weight"):   # head weights: its column
                got, g = got[:, ~kink[m]], g[:, ~kink[m]]
            rep.close_scaled("grad/" + k, got, g, TOL["grad"])
        rep.finish()
    assert seen[0] < seen[1] < seen[2] and seen[3] == seen[0]    # (the needs really differ)


def test_library_refuses_an_undersized_scratch():
    """mopoe_buffers.wgrad_scratch_floats (ABI 11): a step that needs more is an argument
    error before anything is launched, not a write past the buffer."""
    cfg = mo.Config(**C1)
    spec, eng = make_engine(cfg)
    n = 4096
    x = {k: v.cuda() for k, v in mo.make_inputs(cfg.names, cfg.input_dim, n, seed=22).items()}
    plan = spec.plan(list(x), n, backward=True)
    ws = eng.workspace(n, 1, True)
    eng._ensure_scratch(plan, ws)
    buf = eng._buffers(ws, x, None, plan=plan)
    need = L.lib.mopoe_wgrad_scratch_floats(spec.c_model, plan.c_step)
    assert buf.wgrad_scratch_floats == need
    before = eng.params.clone()
    buf.wgrad_scratch_floats = need - 1
    rc = L.lib.mopoe_train_step(spec.c_model, plan.c_step, buf, None, L.stream_ptr())
    assert rc == -1 and b"wgrad_scratch" in L.lib.mopoe_last_error()
    torch.cuda.synchronize()
    assert eng.step_count() == 0 and torch.equal(eng.params, before)    # (nothing was launched)
    buf.wgrad_scratch_floats = need
    L.check(L.lib.mopoe_train_step(spec.c_model, plan.c_step, buf, None, L.stream_ptr()), "train")
    torch.cuda.synchronize()
    eng.check_valid(sync=True)
