"""GPU: the fused launch (encoder layer + per-sample chain in one kernel, with an
in-kernel producer/consumer hand-off of h) against the three-launch form: the same
arithmetic in the same order, so the parameters after a few thousand steps must be
bit-identical -- a hand-off that let a row group read h too early or stale would show
here -- and no hand-off may time out."""
import pytest
import torch

import mopoe_amd as mm

pytestmark = pytest.mark.gpu


def _train(steps, seed, n=256):
    spec = mm.ModelSpec(["clinical", "rois"], [7, 444], [3, 20])
    eng = mm.MoPoEEngine(spec, "cuda", seed=seed)
    eng.reset_parameters(torch.Generator().manual_seed(0))
    g = torch.Generator().manual_seed(1)
    pool = [{"clinical": torch.randn(n, 7, generator=g).cuda(),
             "rois": torch.randn(n, 444, generator=g).cuda()} for _ in range(8)]
    for i in range(steps):
        eng.train_step(pool[i % 8])
    torch.cuda.synchronize()
    assert int(eng.counters[2]) == 0, "a hand-off timed out"
    assert eng.step_count() == steps
    return eng.params.clone(), eng.exp_avg_sq.clone()


@pytest.mark.parametrize("n", [256, 100, 16, 1024])
def test_fused_launch_is_bit_identical_to_three_launches(n, monkeypatch):
    monkeypatch.delenv("MOPOE_NO_FUSE", raising=False)
    fused = _train(3000, 5, n)
    again = _train(3000, 5, n)
    monkeypatch.setenv("MOPOE_NO_FUSE", "1")
    plain = _train(3000, 5, n)
    for a, b, c in zip(fused, again, plain):
        assert torch.equal(a, b)
        assert torch.equal(a, c)
