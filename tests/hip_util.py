"""Shared pieces of the GPU parity tests (HIP path vs oracle / fixtures)."""
from collections import OrderedDict

import torch

import mopoe_oracle as mo


class Report:
    """Collects every mismatch of a case, then fails once with the full list."""

    def __init__(self, title):
        self.title = title
        self.rows = []
        self.bad = []

    def close(self, name, got, want, rtol, atol):
        g = torch.as_tensor(got).detach().cpu().double()
        w = torch.as_tensor(want).detach().cpu().double()
        if g.shape != w.shape:
            if g.numel() == w.numel():
                g = g.reshape(w.shape)
            else:
                self.bad.append("%s: shape %s vs %s" % (name, tuple(g.shape),
                                                        tuple(w.shape)))
                return
        if g.numel() == 0:
            return
        err = (g - w).abs()
        tol = atol + rtol * w.abs()
        nan = not bool(torch.isfinite(g).all())
        worst = float((err / tol).max()) if not nan else float("inf")
        self.rows.append((name, float(err.max()) if not nan else float("nan"),
                          worst))
        if nan or worst > 1.0:
            i = int((err / tol).argmax()) if not nan else 0
            self.bad.append(
                "%s: max|err| %.3e = %.1fx tol (rtol %g atol %g); worst got "
                "%.8g want %.8g; %d/%d out" % (
                    name, float(err.max()), worst, rtol, atol,
                    float(g.reshape(-1)[i]), float(w.reshape(-1)[i]),
                    int((err > tol).sum()), g.numel()))

    def close_scaled(self, name, got, want, scale_tol, rtol=0.0):
        """|got - want| <= scale_tol * max|want| + rtol * |want|: the bound for
        a float32 quantity that is a long signed sum (a gradient), whose
        rounding error scales with the terms, not with the cancelled result."""
        w = torch.as_tensor(want).detach().cpu().double()
        atol = scale_tol * float(w.abs().max()) if w.numel() else 0.0
        self.close(name, got, want, rtol, max(atol, 1e-12))

    def finish(self):
        if self.bad:
            raise AssertionError("%s: %d mismatching tensors\n  %s" % (
                self.title, len(self.bad), "\n  ".join(self.bad)))


def make_engine(cfg, device="cuda"):
    import mopoe_amd as mm
    spec = mm.ModelSpec(cfg.names, cfg.input_dim, cfg.style_dim,
                        class_dim=cfg.class_dim, method=cfg.method,
                        factorized=cfg.factorized, beta=cfg.beta,
                        beta_style=cfg.beta_style, beta_content=cfg.beta_content,
                        initial_out_logvar=cfg.initial_out_logvar,
                        learn_output_scale=cfg.learn_output_scale, lr=cfg.lr,
                        betas=cfg.betas, adam_eps=cfg.adam_eps,
                        poe_unimodal_elbos=cfg.poe_unimodal_elbos,
                        likelihood=cfg.likelihood, enc_layers=cfg.enc_layers,
                        dec_layers=cfg.dec_layers, dropout=cfg.dropout,
                        sample_scale=cfg.sample_scale,
                        gemm_operands=getattr(cfg, "gemm_operands", "f32"))
    eng = mm.MoPoEEngine(spec, device)
    eng.load_params(mo.init_params(cfg, 0))
    return spec, eng


# fp32 tolerances of the HIP path against the float32 CPU oracle.  The MFMA
# f32 path is an exact fmaf chain; differences come from summation order
# (torch/MKL blocked GEMMs and pairwise reductions vs k-ordered chains) and
# from expf/logf implementations (<= 2 ulp).
TOL = dict(
    latent=(2e-5, 2e-5),     # posterior mu / logvar, subsets, joint
    loc=(2e-5, 5e-5),        # reconstructions
    scalar=(2e-5, 1e-5),     # loss terms (values up to ~1e4)
    grad=4e-6,               # gradients: x max|g| of the tensor (the float32
                             # oracle itself sits 4e-7..1e-6 x max|g| from a
                             # float64 evaluation, tests/test_oracle_precision.py)
    moment2=2e-5,            # Adam exp_avg_sq: x max of the tensor
    param1=(1e-5, 5e-6),     # parameters after one Adam step (lr 2e-3: a
                             # sign-level disagreement on a ~0 gradient moves
                             # a weight by up to 4e-3, handled separately)
)


def compare_forward(rep, spec, eng, plan, ws, out, prefix="", check_scale=True):
    """HIP results vs the oracle's basic_routine_epoch / forward output."""
    res_o = out["results"] if "results" in out else out
    lat_o = res_o["latents"]
    res = eng.results(plan, ws)
    lat = res["latents"]
    rt, at = TOL["latent"]
    for k, (mu, lv) in lat_o["modalities"].items():
        g = lat["modalities"][k]
        if mu is None:
            assert g[0] is None and g[1] is None, k
            continue
        rep.close(prefix + "modalities/%s/mu" % k, g[0], mu, rt, at)
        rep.close(prefix + "modalities/%s/logvar" % k, g[1], lv, rt, at)
    assert list(lat["subsets"].keys()) == list(lat_o["subsets"].keys())
    for k, (mu, lv) in lat_o["subsets"].items():
        rep.close(prefix + "subsets/%s/mu" % k, lat["subsets"][k][0], mu, rt, at)
        rep.close(prefix + "subsets/%s/logvar" % k, lat["subsets"][k][1], lv, rt, at)
    rep.close(prefix + "mus", lat["mus"], lat_o["mus"], rt, at)
    rep.close(prefix + "logvars", lat["logvars"], lat_o["logvars"], rt, at)
    rep.close(prefix + "weights", lat["weights"], lat_o["weights"], 1e-7, 0)
    rep.close(prefix + "joint/mu", lat["joint"][0], lat_o["joint"][0], rt, at)
    rep.close(prefix + "joint/logvar", lat["joint"][1], lat_o["joint"][1], rt, at)
    rt, at = TOL["loc"]
    assert list(res["rec"].keys()) == list(res_o["rec"].keys())
    for k, (loc, scale) in res_o["rec"].items():
        rep.close(prefix + "rec/%s/loc" % k, res["rec"][k].loc, loc, rt, at)
        if check_scale:  # (after a fused Adam step the logvar has moved on)
            rep.close(prefix + "rec/%s/scale" % k, res["rec"][k].scale,
                      scale.expand_as(loc), 1e-6, 1e-7)
    rt, at = TOL["scalar"]
    rep.close(prefix + "joint_divergence", res["joint_divergence"],
              res_o["joint_divergence"], rt, at)
    rep.close(prefix + "individual_divs", res["individual_divs"],
              res_o["individual_divs"], rt, at)
    if "total_loss" in out:
        sc = eng.scalars(plan, ws)
        rep.close(prefix + "total_loss", sc["total_loss"], out["total_loss"], rt, at)
        assert list(sc["log_probs"].keys()) == list(out["log_probs"].keys())
        for k, v in out["log_probs"].items():
            rep.close(prefix + "log_probs/" + k, sc["log_probs"][k], v, rt, at)
        assert list(sc["klds"].keys()) == list(out["klds"].keys())
        for k, v in out["klds"].items():
            rep.close(prefix + "klds/" + k, sc["klds"][k], v, rt, at)
        for k, v in out["klds_style"].items():
            rep.close(prefix + "klds_style/" + k, sc["klds_style"][k], v, rt, at)
