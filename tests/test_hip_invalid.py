"""GPU: a training step that cannot be completed is never half-applied and never goes
unnoticed (reference run_epochs.py:180-182 applies a step or nothing).

* a row group of the fused launch whose producers do not show up in time raises the
  sticky MOPOE_CTR_INVALID word; the weight-gradient launch then applies NO Adam update
  (parameters, moments and the per-modality step counts stay put), the engine raises
  from its pinned host mirror without synchronising, and `recover()` re-arms it;
* mopoe_adam_step refuses the update when the ranks' batches held different
  modalities (control words of the all-reduced gradient buffer);
* inputs are read through descriptors of exactly the tensor's size: a gather index
  outside the cohort reads a row of zeros instead of faulting."""
import os
from collections import OrderedDict

import pytest
import torch

import mopoe_amd as mm
import mopoe_oracle as mo
from hip_util import make_engine

pytestmark = pytest.mark.gpu
L = mm._lib
CFG = dict(names=["clinical", "rois"], input_dim=[7, 444], style_dim=[3, 20])


def _batch(cfg, n, seed):
    return mo.make_inputs(cfg.names, cfg.input_dim, n, seed=seed)


def _eps(cfg, n, seed):
    g = mo.noise_rng(seed)
    import numpy as np
    shapes = [(n, cfg.class_dim)] + [(n, s) for s in cfg.style_dim]
    return [torch.from_numpy(g.standard_normal(s).astype(np.float32)) for s in shapes]


def _state(eng):
    return [t.clone() for t in (eng.params, eng.exp_avg, eng.exp_avg_sq)], eng.adam_steps()


def test_timed_out_handoff_applies_nothing_and_raises():
    cfg = mo.Config(**CFG)
    _, eng = make_engine(cfg)
    _, twin = make_engine(cfg)          # never sees the failure
    n = 256                             # configs[1]: the fused launch
    b1, b2, bx = _batch(cfg, n, 1), _batch(cfg, n, 2), _batch(cfg, n, 3)
    e1, e2, ex = _eps(cfg, n, 11), _eps(cfg, n, 12), _eps(cfg, n, 13)
    for e in (eng, twin):
        e.train_step(b1, eps=e1)
    torch.cuda.synchronize()
    before, steps_before = _state(eng)
    os.environ["MOPOE_TEST_HANDOFF_SPINS"] = "0"     # every row group gives up at once
    L.reload_knobs()
    try:
        eng.train_step(bx, eps=ex)
        torch.cuda.synchronize()
    finally:
        del os.environ["MOPOE_TEST_HANDOFF_SPINS"]
        L.reload_knobs()
    after, steps_after = _state(eng)
    for a, b in zip(before, after):
        assert torch.equal(a, b)                     # nothing was applied
    assert steps_after == steps_before
    assert int(eng.counters[L.CTR_INVALID]) > 0
    with pytest.raises(L.MopoeError):
        eng.check_valid(sync=True)
    with pytest.raises(L.MopoeError):                # the next step refuses to start,
        eng.train_step(b2, eps=e2)                   # from the pinned host mirror alone
    assert int(eng.status_host[1]) > 0
    # while the word is up, even a forced launch applies nothing
    eng.status_host[1] = 0
    eng.train_step(b2, eps=e2)
    torch.cuda.synchronize()
    for a, b in zip(before, _state(eng)[0]):
        assert torch.equal(a, b)
    eng.recover()
    eng.check_valid(sync=True)
    eng.train_step(b2, eps=e2)
    twin.train_step(b2, eps=e2)
    torch.cuda.synchronize()
    eng.check_valid(sync=True)
    # the recovered engine continues exactly where the last complete step left it
    assert torch.equal(eng.params, twin.params)
    assert torch.equal(eng.exp_avg_sq, twin.exp_avg_sq)
    assert eng.adam_steps() == twin.adam_steps() == OrderedDict(clinical=2, rois=2)


def test_adam_kernel_checks_the_ranks_modalities():
    cfg = mo.Config(**CFG)
    spec, a = make_engine(cfg)
    _, b = make_engine(cfg)
    x, eps = _batch(cfg, 64, 5), _eps(cfg, 64, 6)
    for e in (a, b):
        e.train_step(x, eps=eps, apply_adam=False)
    torch.cuda.synchronize()
    ctrl = spec.c_model.off_ctrl
    assert a.grads[ctrl:ctrl + 2].tolist() == [1.0, 1.0]
    a.adam_step()                                   # one rank
    b.grads.mul_(2.0)                               # "sum over two identical ranks"
    b.adam_step(world=2)
    torch.cuda.synchronize()
    assert torch.equal(a.params, b.params) and torch.equal(a.exp_avg_sq, b.exp_avg_sq)
    assert a.adam_steps() == b.adam_steps() == OrderedDict(clinical=1, rois=1)
    # second step: the other rank's batch lacked `rois`
    x2, eps2 = _batch(cfg, 64, 7), _eps(cfg, 64, 8)
    b.train_step(x2, eps=eps2, apply_adam=False)
    b.grads.mul_(2.0)
    b.grads[ctrl + 1] = 1.0
    before = b.params.clone()
    b.adam_step(world=2)
    torch.cuda.synchronize()
    assert torch.equal(before, b.params)
    assert b.adam_steps() == OrderedDict(clinical=1, rois=1)
    with pytest.raises(L.MopoeError):
        b.check_valid(sync=True)
    assert int(b.status_host[1]) > 0


def test_mixed_masks_keep_torch_per_parameter_steps():
    """Free-running (no re-synchronisation): six steps whose batches hold changing
    modality sets, HIP vs oracle with torch's per-parameter Adam step counts."""
    cfg = mo.Config(**CFG)
    _, eng = make_engine(cfg)
    params = mo.init_params(cfg, 0)
    state = mo.adam_init(params)
    masks = [["clinical", "rois"], ["rois"], ["rois"], ["clinical", "rois"], ["clinical"],
             ["clinical", "rois"]]
    for k, present in enumerate(masks):
        x = OrderedDict((m, v) for m, v in _batch(cfg, 48, 20 + k).items() if m in present)
        noise = mo.Noise(generator=mo.noise_rng(40 + k))
        mo.train_step(params, cfg, x, noise, state)
        eng.train_step(x, eps=noise.tape)
    torch.cuda.synchronize()
    assert eng.adam_steps() == OrderedDict(clinical=4, rois=5)
    got = eng.named_params()
    for k, v in params.items():
        err = (got[k].cpu() - v).abs().max().item()
        assert err < 2e-4, (k, err)     # (sign-level flips of ~0 gradients move a weight by lr)
        assert (got[k].cpu() - v).abs().mean().item() < 2e-6, k


def test_inputs_are_read_through_exact_descriptors():
    cfg = mo.Config(**CFG)
    _, a = make_engine(cfg)
    _, b = make_engine(cfg)
    n = 48
    x = _batch(cfg, n, 9)
    eps = _eps(cfg, n, 10)
    # cohort arrays of exactly n rows; batch rows 3 and 17 point outside the cohort
    idx = torch.arange(n, dtype=torch.int32)
    idx[3], idx[17] = n + 5, 10 ** 6
    zeroed = OrderedDict((k, v.clone()) for k, v in x.items())
    for v in zeroed.values():
        v[3] = 0
        v[17] = 0
    _, ws_a = a.train_step({k: v.cuda() for k, v in x.items()}, eps=eps, row_index=idx)
    _, ws_b = b.train_step(zeroed, eps=eps)
    torch.cuda.synchronize()
    assert torch.equal(ws_a.stats, ws_b.stats)
    assert torch.equal(a.params, b.params)
    # the C ABI refuses a gather without the cohort's row count
    plan = a.spec.plan(list(x), n, backward=True)
    buf = a._buffers(a.workspace(n, 1, True), {k: v.cuda() for k, v in x.items()},
                     {k: idx.cuda() for k in x})
    buf.x_rows[0] = 0
    rc = L.lib.mopoe_train_step(a.spec.c_model, plan.c_step, buf, None, L.stream_ptr())
    assert rc == -1 and b"x_rows" in L.lib.mopoe_last_error()
