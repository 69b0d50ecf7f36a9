"""CPU: how far float32 rounding alone moves the oracle's gradients (float32
vs float64 evaluation of the same restatement).  This calibrates the gradient
tolerance the GPU parity tests use (tests/hip_util.py TOL['grad'])."""
import torch

import mopoe_oracle as mo
from golden_util import Fixture
from hip_util import TOL


def test_float32_oracle_gradient_noise_is_below_the_hip_tolerance():
    worst = 0.0
    for case in ("c1_joint_fact_n32", "c3_poe_fact_n32", "moe_fact_n32",
                 "c5_4mod_joint_fact_n32"):
        fx = Fixture(case)
        cfg = fx.cfg
        p = mo.init_params(cfg, 0)
        x = fx.inputs()
        n = fx.noise(0)
        _, g32 = mo.loss_and_grads(p, cfg, x, n)
        cfg.dtype = torch.float64
        _, g64 = mo.loss_and_grads(p, cfg, x, mo.Noise(tape=n.tape))
        cfg.dtype = torch.float32
        for k in g32:
            rel = ((g32[k].double() - g64[k]).abs().max() /
                   g64[k].abs().max()).item()
            worst = max(worst, rel)
    # float32 noise is a few 1e-7 of the tensor's largest gradient; the HIP
    # tolerance leaves roughly an order of magnitude above it
    assert 1e-8 < worst < TOL["grad"] / 2


def test_what_bfloat16_operands_in_the_encoder_layer_cost():
    """The opt-in of the HIP path (ModelSpec(gemm_operands="bf16"): the encoder layer's GEMM of
    a large batch multiplies bfloat16 roundings, float32 sums) against the float32 step, on
    the oracle: the loss moves in its fifth digit; every gradient behind the first layer by a
    few 1e-3 of its tensor's largest entry -- a thousand times the float32 noise measured
    above; the first layer's own gradient by MORE, because a pre-activation within 2^-9 of
    zero changes sides of the ReLU and takes its row's contribution along (a fifth of the
    largest entry in these 32-row batches, where one row is 3 % of the sum).  That is why the
    default stays float32; the GPU test of the opt-in (tests/test_hip_large_batch.py) holds the
    kernel to THIS definition at the float32 tolerances."""
    worst_loss, worst_first, worst_rest = 0.0, 0.0, 0.0
    for case in ("c1_joint_fact_n32", "c3_poe_fact_n32", "c5_4mod_joint_fact_n32"):
        fx = Fixture(case)
        cfg = fx.cfg
        p = mo.init_params(cfg, 0)
        x = fx.inputs()
        n = fx.noise(0)
        out32, g32 = mo.loss_and_grads(p, cfg, x, n)
        cfg.gemm_operands = "bf16"
        out16, g16 = mo.loss_and_grads(p, cfg, x, mo.Noise(tape=n.tape))
        cfg.gemm_operands = "f32"
        worst_loss = max(worst_loss, abs(float(out16["total_loss"]) - float(out32["total_loss"]))
                         / abs(float(out32["total_loss"])))
        for k in g32:
            rel = ((g16[k] - g32[k]).abs().max() / g32[k].abs().max()).item()
            if ".shared_encoder.0." in k:
                worst_first = max(worst_first, rel)
            else:
                worst_rest = max(worst_rest, rel)
    print("bf16 operands: loss %.2e relative; gradients %.2e (first layer: %.2e) of their largest "
          "entry" % (worst_loss, worst_rest, worst_first))
    assert 1e-7 < worst_loss < 2e-4
    assert 1e-4 < worst_rest < 1e-2
    assert worst_first < 0.5
