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
