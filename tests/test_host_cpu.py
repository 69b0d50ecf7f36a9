"""CPU: the host side of the boundary -- the C ABI loads and exports what
include/mopoe_hip.h declares, descriptors are validated before any HIP call,
the mirror modules keep the reference's state_dict keys, and the per-batch
plan reproduces the reference's float32 slice arithmetic."""
import ctypes as C
import os
import re

import pytest
import torch

import mopoe_amd as mm
import mopoe_oracle as mo
from golden_util import Fixture, case_names
from surface_util import make_experiment

L = mm._lib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "mopoe_hip.h")).read()
    declared = set(re.findall(r"^(?:int|int64_t|const char\*)\s+(mopoe_[a-z0-9_]+)\s*\(", header, re.M))
    assert declared, "no declarations parsed"
    assert declared == set(L.SYMBOLS), declared ^ set(L.SYMBOLS)
    for name in declared:
        assert getattr(L.lib, name) is not None
    assert L.lib.mopoe_abi_version() == L.ABI_VERSION


def test_struct_mirrors_match_the_c_layout():
    sizes = [C.sizeof(L.Model), C.sizeof(L.Step), C.sizeof(L.Buffers), C.sizeof(L.Adam)]
    assert [L.lib.mopoe_sizeof(i) for i in range(4)] == sizes
    assert L.lib.mopoe_sizeof(99) == -1


def test_descriptors_are_rejected_before_any_gpu_work():
    spec = mm.ModelSpec(["clinical", "rois"], [7, 444], [3, 20])
    plan = spec.plan(["clinical", "rois"], 32, backward=True)
    buf = L.Buffers()                       # all NULL
    rc = L.lib.mopoe_train_step(spec.c_model, plan.c_step, buf, None, None)
    assert rc == -1 and b"null" in L.lib.mopoe_last_error()
    bad = L.Step.from_buffer_copy(plan.c_step)
    bad.present_mask = 0
    assert L.lib.mopoe_forward(spec.c_model, bad, buf, None) == -1
    assert b"present_mask" in L.lib.mopoe_last_error()
    with pytest.raises(L.MopoeError):
        L.check(-1, "x")
    assert L.lib.mopoe_poe(None, None, 1, 1, 1e-8, None, None, None) == -1
    bad = L.Step.from_buffer_copy(plan.c_step)
    bad.likelihood = 7
    assert L.lib.mopoe_forward(spec.c_model, bad, buf, None) == -1
    assert b"likelihood" in L.lib.mopoe_last_error()


def test_likelihood_of_the_flags_reaches_the_step():
    """modalities/modality.py:18-30: the decoder's (loc, scale) pair parameterises Normal or
    Laplace; the other names of the reference's switch take other arguments."""
    import types
    flags = types.SimpleNamespace(
        input_dim=[7, 444], style_dim=[3, 20], class_dim=20, modality_poe=False,
        modality_moe=False, factorized_representation=True, beta=1.0, beta_style=1.0,
        beta_content=1.0, initial_out_logvar=-3.0, learn_output_scale=True,
        likelihood="laplace")
    spec = mm.ModelSpec.from_flags(flags, ["clinical", "rois"])
    assert spec.likelihood == "laplace"
    assert spec.plan(["clinical", "rois"], 32, backward=True).c_step.likelihood == L.LIKELIHOODS["laplace"]
    assert mm.ModelSpec(["a"], [5], [2]).plan(["a"], 8).c_step.likelihood == L.LIKELIHOODS["normal"]
    flags.likelihood = "bernoulli"
    with pytest.raises(NotImplementedError):
        mm.ModelSpec.from_flags(flags, ["clinical", "rois"])


def test_no_cpu_fallback():
    spec = mm.ModelSpec(["clinical", "rois"], [7, 444], [3, 20])
    eng = mm.MoPoEEngine(spec, "cpu")
    x = mo.make_inputs(spec.names, spec.input_dim, 8, seed=0)
    if not torch.cuda.is_available():
        with pytest.raises(L.MopoeError):
            eng.train_step(x)
        with pytest.raises(L.MopoeError):
            eng.forward(x)


def test_flat_layout_covers_every_reference_parameter():
    for names, dims, style, fact in ((["clinical", "rois"], [7, 444], [3, 20], True),
                                     (["a", "b", "c", "d"], [7, 444, 128, 64], [3, 20], True),
                                     (["clinical", "rois"], [7, 444], [3, 20], False)):
        cfg = mo.Config(names, dims, style, factorized=fact)
        spec = mm.ModelSpec(names, dims, style, factorized=fact)
        flat = torch.arange(spec.num_floats, dtype=torch.float32)
        views = spec.param_views(flat)
        shapes = mo.param_shapes(cfg)
        assert set(views) == set(shapes)
        seen = torch.zeros(spec.num_floats, dtype=torch.int32)
        for k, v in views.items():
            assert tuple(v.shape) == tuple(shapes[k]), k
            seen[v.reshape(-1).long()] += 1
        assert int(seen.max()) == 1                      # no overlap
        assert int(seen.sum()) == sum(v.numel() for v in views.values())
        c = spec.c_model
        for m in range(len(names)):                      # 64-float alignment
            for off in (c.off_w1[m], c.off_b1[m], c.off_wh[m], c.off_bh[m],
                        c.off_wd[m], c.off_bd[m], c.off_lvo[m]):
                assert off % 64 == 0


@pytest.mark.parametrize("method", ["joint_elbo", "poe", "moe"])
def test_model_keeps_reference_state_dict_keys(method):
    cfg = mo.Config(["clinical", "rois"], [7, 444], [3, 20], method=method)
    exp = make_experiment(cfg, "cpu")
    model = exp.models
    init = mo.init_params(cfg, 0)
    assert set(model.state_dict().keys()) == set(init.keys())
    missing, unexpected = model.load_state_dict(init, strict=True)
    assert not missing and not unexpected
    base = model.engine.params
    for k, p in model.named_parameters():
        assert torch.equal(p.detach(), init[k]), k
        assert base.data_ptr() <= p.data_ptr() < base.data_ptr() + 4 * base.numel()
    assert [k for k, p in model.named_parameters() if not p.requires_grad] == []
    assert list(exp.subsets.keys()) == ["", "clinical", "rois", "clinical_rois"]


@pytest.mark.parametrize("case", case_names())
def test_plan_reproduces_reference_mixture_and_subsets(case):
    fx = Fixture(case)
    cfg = fx.cfg
    spec = mm.ModelSpec(cfg.names, cfg.input_dim, cfg.style_dim, method=cfg.method,
                        factorized=cfg.factorized)
    plan = spec.plan(fx.present, fx.N, backward=True)
    # subset keys / availability as the reference recorded them
    want = [k.split("/")[2] for k in fx.keys("step0/klds/")]
    assert plan.avail_keys == want
    K = len(fx.z["step0/weights"])
    assert plan.c_step.num_comp == K
    w = mo.reweight_weights((1 / float(K)) * torch.ones(K))
    starts, ends = mo.mixture_bounds(fx.N, w)
    f = plan.c_step.comp_f
    for n in range(fx.N):
        k = min(n // f, K - 1) if f > 0 else K - 1
        assert starts[k] <= n < ends[k]
    assert abs(sum(plan.comp_w) - 1.0) < 1e-6
    assert plan.lds_bytes() <= 160 * 1024


def test_group_rows_sizes_the_row_position_rules_for_one_repeat():
    """f2: with group_rows the mixture / moe slice sizes are those of ONE repeat
    (the batch mixture_component_selection sees in the reference's separate
    forwards, utils/utils.py:63-85), not of the folded batch."""
    spec = mm.ModelSpec(["clinical", "rois"], [7, 444], [3, 20], method="joint_elbo")
    names = ["clinical", "rois"]
    one = spec.plan(names, 50)
    folded = spec.plan(names, 50 * 40, group_rows=50)
    assert folded.c_step.group_rows == 50 and folded.c_step.n == 2000
    assert folded.c_step.comp_f == one.c_step.comp_f == 16   # floor(50/3)
    assert spec.plan(names, 2000).c_step.comp_f == 666
    moe = mm.ModelSpec(names, [7, 444], [3, 20], method="moe")
    a, b = moe.plan(names, 50), moe.plan(names, 500, group_rows=50)
    assert list(a.c_step.sub_f) == list(b.c_step.sub_f)
    with pytest.raises(ValueError):
        spec.plan(names, 2000, group_rows=30)          # must divide n
    with pytest.raises(ValueError):
        spec.plan(names, 2000, backward=True, group_rows=50)   # forward-only


def test_input_matrices_are_passed_as_they_are():
    """ABI 6: x[m] is read through a descriptor of exactly x_rows[m] * d_m floats, so the
    caller owes no slack after the last row and the host side makes no copy of a matrix
    that is already float32, contiguous and on the device."""
    L = mm._lib
    assert not hasattr(L, "rows_with_slack")
    flush = torch.randn(256, 7)                       # storage ends with the last row
    assert L.device_rows(flush, "cpu") is flush
    tail = torch.randn(300, 7)[:256]
    assert L.device_rows(tail, "cpu") is tail
    assert L.device_rows(flush.double(), "cpu").dtype == torch.float32
    assert L.device_rows(torch.randn(7, 256).t(), "cpu").is_contiguous()


def test_hot_kernels_keep_their_registers_and_scratch():
    """The step's speed rests on compiler settings (csrc/Makefile: -Os, iterative-ILP
    scheduler) whose effect is visible in the per-kernel resource report the build
    keeps next to the library: any sizeable private segment cost 3x on MI355X, and a
    1024-thread workgroup cannot hold more than 128 VGPRs per lane.  A toolchain
    change that moves these fails here instead of silently costing microseconds."""
    path = os.path.join(os.path.dirname(L.LIB_PATH), "csrc", "resources.txt")
    assert os.path.exists(path), "build with make -C .../csrc (it writes resources.txt)"
    assert os.path.getmtime(path) >= os.path.getmtime(L.LIB_PATH) - 120
    text = open(path).read()
    kernels = {}
    for block in text.split("Function Name: ")[1:]:
        name = block.split()[0]
        get = lambda key: int(re.search(re.escape(key) + r": (\d+)", block).group(1))
        kernels[name] = dict(vgpr=get("VGPRs"), scratch=get("ScratchSize [bytes/lane]"),
                             sgpr_spill=get("SGPRs Spill"), occupancy=get("Occupancy [waves/SIMD]"))
    pick = lambda frag: next(v for k, v in kernels.items() if frag in k)
    lean, generic = pick("k_fusedILi1EE"), pick("k_fusedILi0EE")
    latent = pick("k_latentILi0EE")
    # (measured builds of the lean launch: 24..80 bytes of scratch all ran within 1 % of each
    #  other; 112 bytes and more cost a microsecond)
    assert lean["vgpr"] <= 128 and lean["scratch"] <= 96 and lean["occupancy"] >= 4, lean
    assert generic["vgpr"] <= 128 and generic["scratch"] <= 128, generic
    quad = pick("k_fusedILi4EE")                         # the four-row form (the headline's launch)
    assert quad["vgpr"] <= 128 and quad["scratch"] <= 32 and quad["occupancy"] >= 4, quad
    quad2 = pick("k_fusedILi5EE")                        # ... with method poe's second decoder pass
    assert quad2["vgpr"] <= 128 and quad2["scratch"] <= 32 and quad2["occupancy"] >= 4, quad2
    for frag in ("k_fusedILi2EE", "k_fusedILi3EE"):      # the poe / four-modality forms
        assert pick(frag)["vgpr"] <= 128 and pick(frag)["scratch"] <= 256, (frag, pick(frag))
    assert latent["vgpr"] <= 128 and latent["scratch"] == 0, latent
    for frag in ("k_latentILi1EE", "k_latentILi2EE", "k_latentILi3EE"):   # large batches: the specialised bodies
        assert pick(frag)["vgpr"] <= 128 and pick(frag)["scratch"] <= 64, (frag, pick(frag))
    # the separate encoder-layer launch: an instantiation per K split, TWO waves per SIMD (as one
    # kernel without the bound it took 255 VGPRs + 40 AGPRs: one wave per SIMD, every launch in
    # two rounds -- DESIGN.md section 5.5)
    for frag in ("8k_linearILi1EE", "8k_linearILi2EE", "8k_linearILi4EE"):
        assert pick(frag)["occupancy"] >= 2 and pick(frag)["scratch"] <= 16, (frag, pick(frag))
    assert pick("k_wgrad_bigE")["occupancy"] >= 7 and pick("k_wgrad_bigE")["scratch"] == 0
    for frag in ("k_wgradILi4ELb0", "k_wgradILi8ELb0", "k_wgradILi4ELb1", "k_wgradILi8ELb1",
                 "k_adamE", "k_xgmiE", "k_linear_bigILi64ELb0EE", "k_linear_bigILi64ELb1EE"):
        assert pick(frag)["scratch"] == 0, (frag, pick(frag))
