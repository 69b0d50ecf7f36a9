"""GPU: the fragment-major weight copies (mopoe_buffers.wfrag) the four-row form of the
fused launch reads.  The contract of include/mopoe_hip.h: every update the library
applies keeps them in step with `params`; after any OTHER write mopoe_wfrag_refresh must
run -- the engine does that by itself when torch has counted a write to the flat buffer
(tensor._version), and exposes refresh_wfrag() for writes torch does not count."""
import pytest
import torch

import mopoe_amd as mm

pytestmark = pytest.mark.gpu

NAMES, DIMS, STYLE = ["clinical", "rois"], [7, 444], [3, 20]


def _pool(n=64, k=4):
    g = torch.Generator().manual_seed(3)
    return [{"clinical": torch.randn(n, 7, generator=g).cuda(),
             "rois": torch.randn(n, 444, generator=g).cuda()} for _ in range(k)]


def _engine(seed=4):
    eng = mm.MoPoEEngine(mm.ModelSpec(NAMES, DIMS, STYLE), "cuda", seed=seed)
    eng.reset_parameters(torch.Generator().manual_seed(0))
    return eng


def _expected_copies(eng):
    """WF[tile][k/4][lane][4] = W[64 tile + lane][k .. k+3] of the heads and decoder
    weights (csrc/mopoe_common.h: WFrag), rebuilt on the host from the parameters."""
    out = torch.zeros_like(eng.wfrag)
    off = 0
    for name in NAMES:
        for key, pad in (("encoders.%s.", None), ("decoders.%s.out_mu.weight", 4)):
            if pad is None:
                w = torch.cat([eng.views["encoders.%s.%s.weight" % (name, h)]
                               for h in ("style_mu", "style_logvar", "class_mu", "class_logvar")])
            else:
                w = eng.views[key % name]
            rows, k = w.shape
            k4 = (k + 3) // 4
            tiles = (rows + 63) // 64
            wp = torch.zeros(tiles * 64, k4 * 4, device=w.device)
            wp[:rows, :k] = w
            blk = wp.view(tiles, 64, k4, 4).permute(0, 2, 1, 3).reshape(-1)
            out[off:off + blk.numel()] = blk
            off += blk.numel()
    assert off == out.numel()
    return out


def test_copies_follow_every_update_of_the_library():
    eng = _engine()
    pool = _pool()
    for i in range(7):                       # fused Adam epilogue of the weight-gradient launch
        eng.train_step(pool[i % 4])
    torch.cuda.synchronize()
    assert torch.equal(eng.wfrag, _expected_copies(eng))
    for i in range(3):                       # separate update: k_adam, then the rebuild kernel
        eng.train_step(pool[i % 4], apply_adam=False)
        eng.adam_step()
    torch.cuda.synchronize()
    assert torch.equal(eng.wfrag, _expected_copies(eng))
    # a batch without a modality leaves that modality's weights and copies alone
    eng.train_step({"rois": pool[0]["rois"]})
    torch.cuda.synchronize()
    assert torch.equal(eng.wfrag, _expected_copies(eng))


def test_an_outside_write_is_followed_by_a_refresh():
    pool = _pool()
    a, b = _engine(seed=4), _engine(seed=4)
    for i in range(3):
        a.train_step(pool[i % 4])
    # a: parameters, moments and step counts of a fresh engine b, written from Python
    with torch.no_grad():
        for k, v in b.views.items():
            a.views[k].copy_(v)              # (views share the flat buffer's version counter)
        a.exp_avg.zero_()
        a.exp_avg_sq.zero_()
        a.counters.copy_(b.counters)
    for i in range(5):
        a.train_step(pool[i % 4])
        b.train_step(pool[i % 4])
    torch.cuda.synchronize()
    assert torch.equal(a.params, b.params)
    assert torch.equal(a.wfrag, _expected_copies(a))


def test_refresh_after_a_write_torch_does_not_count():
    pool = _pool()
    a, b = _engine(seed=4), _engine(seed=4)
    a.train_step(pool[0])
    b.train_step(pool[0])
    scale = 1.0 + 1e-3
    a.params.data.mul_(scale)                # `.data`: its own version counter
    a.refresh_wfrag()
    with torch.no_grad():
        b.params.mul_(scale)
    for i in range(4):
        a.train_step(pool[i % 4])
        b.train_step(pool[i % 4])
    torch.cuda.synchronize()
    assert torch.equal(a.params, b.params)


def test_the_c_entry_point_refuses_null_buffers():
    from importlib import import_module
    L = import_module("2022_cambroise_interpret_multivae_amd._lib")
    spec = mm.ModelSpec(NAMES, DIMS, STYLE)
    rc = L.lib.mopoe_wfrag_refresh(spec.c_model, L.Buffers(), None)
    assert rc != 0 and b"null" in L.lib.mopoe_last_error()
