"""GPU: the fused launch (encoder layer + per-sample chain in one kernel, with an
in-kernel producer/consumer hand-off of h) against the three-launch form: the same
arithmetic in the same order, so the parameters after a few thousand steps must be
bit-identical -- a hand-off that let a row group read h too early or stale would show
here -- and no hand-off may time out."""
import pytest
import torch

import mopoe_amd as mm

pytestmark = pytest.mark.gpu


def _train(steps, seed, n=256, method="joint_elbo"):
    spec = mm.ModelSpec(["clinical", "rois"], [7, 444], [3, 20], method=method)
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
    """(The 16-row groups: MOPOE_QUAD=0.  The four-row groups of small batches add the K
    parts of their GEMMs in another order: next test.)"""
    monkeypatch.delenv("MOPOE_NO_FUSE", raising=False)
    monkeypatch.setenv("MOPOE_QUAD", "0")
    fused = _train(3000, 5, n)
    again = _train(3000, 5, n)
    monkeypatch.setenv("MOPOE_NO_FUSE", "1")
    plain = _train(3000, 5, n)
    for a, b, c in zip(fused, again, plain):
        assert torch.equal(a, b)
        assert torch.equal(a, c)


@pytest.mark.parametrize("n,method", [(256, "joint_elbo"), (100, "joint_elbo"), (16, "joint_elbo"),
                                      (5, "joint_elbo"), (4, "joint_elbo"),
                                      (256, "poe"), (37, "poe"), (512, "joint_elbo"), (400, "poe"),
                                      (768, "joint_elbo"), (1024, "poe"), (640, "poe")])
def test_four_row_groups_are_deterministic_and_track_the_three_launches(n, method, monkeypatch):
    """The four-row form (<= 2 modalities, up to 1,024 rows; beyond 512 rows its launch has more
    blocks than CUs -- the producers first, the row groups on the CUs they leave) is the same
    step with other summation orders: bit-identical from run to run
    (3000 steps, no hand-off times out, the fragment-major weight copies follow every update;
    the two LDS adds per element of its fusion backward commute), and within float32 rounding
    of the three-launch form while rounding has had no time to grow."""
    monkeypatch.delenv("MOPOE_NO_FUSE", raising=False)
    monkeypatch.delenv("MOPOE_QUAD", raising=False)
    quad = _train(3000, 5, n, method)
    again = _train(3000, 5, n, method)
    for a, b in zip(quad, again):
        assert torch.equal(a, b)
    short = _train(20, 5, n, method)
    monkeypatch.setenv("MOPOE_QUAD", "0")
    wide = _train(3000, 5, n, method)
    assert not torch.equal(quad[0], wide[0])                # (the form under test did run)
    monkeypatch.setenv("MOPOE_NO_FUSE", "1")
    plain = _train(20, 5, n, method)
    # 20 Adam steps of lr 1e-3 move a parameter by <= 2e-2; rounding-level differences of
    # the gradients move the two forms apart by a small fraction of that
    assert (short[0] - plain[0]).abs().max().item() < 2e-4
    assert (short[1] - plain[1]).abs().max().item() < 1e-3 * plain[1].abs().max().item()


def test_four_row_groups_behind_an_encoder_layer_launch_of_its_own(monkeypatch):
    """MOPOE_QUAD_OVERSUB=0: beyond 512 rows the encoder layer runs as a launch of its own and
    the fused launch holds row groups only (no producers: the hand-off word is never raised
    and never waited for).  Deterministic, and within float32 rounding of the three launches."""
    monkeypatch.delenv("MOPOE_NO_FUSE", raising=False)
    monkeypatch.delenv("MOPOE_QUAD", raising=False)
    monkeypatch.setenv("MOPOE_QUAD_OVERSUB", "0")
    split = _train(1000, 5, 768, "poe")
    again = _train(1000, 5, 768, "poe")
    for a, b in zip(split, again):
        assert torch.equal(a, b)
    short = _train(20, 5, 768, "poe")
    monkeypatch.delenv("MOPOE_QUAD_OVERSUB")
    one_launch = _train(1000, 5, 768, "poe")
    assert not torch.equal(split[0], one_launch[0])         # (another order of the encoder layer's sums)
    monkeypatch.setenv("MOPOE_QUAD", "0")
    monkeypatch.setenv("MOPOE_NO_FUSE", "1")
    plain = _train(20, 5, 768, "poe")
    assert (short[0] - plain[0]).abs().max().item() < 2e-4
    assert (short[1] - plain[1]).abs().max().item() < 1e-3 * plain[1].abs().max().item()


SHAPES = {
    # BASELINE configs[2]: method poe -> the two-pass form of the fused launch
    "C3_poe_bs1024": dict(names=["clinical", "rois"], dims=[7, 444], style=[3, 20], method="poe",
                          n=1024, steps=300),
    # configs[4]: four modalities, 15 subsets -> the four-modality form
    "C5_four_mods_bs512": dict(names=["a", "b", "c", "d"], dims=[7, 444, 128, 64],
                               style=[3, 3, 3, 3], method="joint_elbo", n=512, steps=300),
    "poe_nofact_bs256": dict(names=["clinical", "rois"], dims=[7, 444], style=[3, 20],
                             method="poe", n=256, steps=300, factorized=False),
}


@pytest.mark.parametrize("shape", sorted(SHAPES))
def test_specialised_forms_are_bit_identical_to_the_generic_one(shape, monkeypatch):
    """The instantiations of the fused launch (latent_body's FORM: constant trip counts,
    narrower expert arrays, compiled-out paths) must change nothing but the time:
    MOPOE_NO_LEAN=1 runs the generic form, MOPOE_NO_FUSE=1 the three launches."""
    c = SHAPES[shape]

    def train():
        spec = mm.ModelSpec(c["names"], c["dims"], c["style"], method=c["method"],
                            factorized=c.get("factorized", True))
        eng = mm.MoPoEEngine(spec, "cuda", seed=9)
        eng.reset_parameters(torch.Generator().manual_seed(0))
        g = torch.Generator().manual_seed(2)
        pool = [{k: torch.randn(c["n"], d, generator=g).cuda()
                 for k, d in zip(c["names"], c["dims"])} for _ in range(4)]
        for i in range(c["steps"]):
            eng.train_step(pool[i % 4])
        eng.check_valid(sync=True)
        return eng.params.clone(), eng.exp_avg_sq.clone()

    for v in ("MOPOE_NO_LEAN", "MOPOE_NO_FUSE"):
        monkeypatch.delenv(v, raising=False)
    monkeypatch.setenv("MOPOE_QUAD", "0")   # (the 16-row instantiations; the four-row ones: above)
    special = train()
    monkeypatch.setenv("MOPOE_NO_LEAN", "1")
    generic = train()
    monkeypatch.delenv("MOPOE_NO_LEAN")
    monkeypatch.setenv("MOPOE_NO_FUSE", "1")
    plain = train()
    for a, b, d in zip(special, generic, plain):
        assert torch.equal(a, b)
        assert torch.equal(a, d)


def _forms_of(steps, n, method):
    """Launch counts per kernel over `steps` training steps (the library's event log)."""
    L = mm._lib
    spec = mm.ModelSpec(["clinical", "rois"], [7, 444], [3, 20], method=method)
    eng = mm.MoPoEEngine(spec, "cuda", seed=5)
    eng.reset_parameters(torch.Generator().manual_seed(0))
    g = torch.Generator().manual_seed(1)
    x = {"clinical": torch.randn(n, 7, generator=g).cuda(), "rois": torch.randn(n, 444, generator=g).cuda()}
    L.profile_enable(True)
    try:
        for _ in range(steps):
            eng.train_step(x)
        torch.cuda.synchronize()
        prof = L.profile_read()
    finally:
        L.profile_enable(False)
    eng.check_valid(sync=True)
    return {k: v[0] for k, v in prof.items()}


def test_oversubscribed_launch_is_held_to_the_device(monkeypatch):
    """The launch of more workgroups than compute units (row groups spinning on producers of the
    same grid) is only taken while every producer is resident first -- counted on the device the
    call runs on (its CU count x the runtime's occupancy for the kernel; MOPOE_FUSE_BLOCKS stands
    in for a smaller device here), not on the constant 256.  Method poe, 768 rows, four K parts =
    240 producers + 192 row groups: on 256 CUs one launch; told that 200 workgroups fit, the
    encoder layer must run as a launch of its own in front of the row groups -- the form
    MOPOE_QUAD_OVERSUB=0 selects, bit for bit."""
    monkeypatch.delenv("MOPOE_NO_FUSE", raising=False)
    monkeypatch.delenv("MOPOE_QUAD", raising=False)
    monkeypatch.setenv("MOPOE_QUAD_OVERSUB", "4")
    whole = _forms_of(4, 768, "poe")
    assert whole["k_fused"] == 4 and whole["k_linear"] == 0        # one launch, producers inside
    monkeypatch.setenv("MOPOE_FUSE_BLOCKS", "200")
    held = _forms_of(4, 768, "poe")
    assert held["k_fused"] == 4 and held["k_linear"] == 4          # the fallback form ran
    guarded = _train(300, 5, 768, "poe")
    monkeypatch.delenv("MOPOE_FUSE_BLOCKS")
    monkeypatch.setenv("MOPOE_QUAD_OVERSUB", "0")
    split = _train(300, 5, 768, "poe")
    for a, b in zip(guarded, split):
        assert torch.equal(a, b)
    # fewer "CUs" than row groups: no four-row form at all, and nothing fused that does not fit
    monkeypatch.setenv("MOPOE_FUSE_BLOCKS", "100")
    monkeypatch.setenv("MOPOE_QUAD_OVERSUB", "4")
    small = _forms_of(2, 768, "poe")
    assert small["k_fused"] == 0 and small["k_latent"] == 2 and small["k_linear"] == 2
