"""Thin tensor-level wrappers over the free functions of the C ABI (float32
device tensors in, float32 device tensors out, no autograd graph).  These back
the reference-named modules divergence_measures/, utils/utils.py and the
standalone Encoder / Decoder forward."""
import torch

from . import _lib as L


def _f32(t):
    L.require_gpu(t)
    return t.detach().to(torch.float32).contiguous()


def linear(x, weight, bias=None, relu=False):
    """y = act(x @ weight^T + bias) through mopoe_linear (MFMA kernel)."""
    x, weight = _f32(x), _f32(weight)
    bias = None if bias is None else _f32(bias)
    n, k = x.shape
    ncols = weight.shape[0]
    if weight.shape[1] != k:
        raise ValueError("linear: x is (N,%d) but weight is %s" % (k, tuple(weight.shape)))
    y = torch.empty(n, ncols, dtype=torch.float32, device=x.device)
    L.check(L.lib.mopoe_linear(L.ptr(x), n, k, L.ptr(weight), L.ptr(bias), ncols,
                               int(bool(relu)), L.ptr(y), L.stream_ptr()), "mopoe_linear")
    return y


def poe(mu, logvar, eps=1e-8):
    mu, logvar = _f32(mu), _f32(logvar)
    if mu.shape != logvar.shape or mu.dim() < 2:
        raise ValueError("poe: mu/logvar must share a shape (E, ...)")
    e = mu.shape[0]
    numel = mu[0].numel()
    out_mu = torch.empty(mu.shape[1:], dtype=torch.float32, device=mu.device)
    out_lv = torch.empty_like(out_mu)
    L.check(L.lib.mopoe_poe(L.ptr(mu), L.ptr(logvar), e, numel, float(eps),
                            L.ptr(out_mu), L.ptr(out_lv), L.stream_ptr()), "mopoe_poe")
    return out_mu, out_lv


def kl_divergence(mu, logvar, norm_value=None):
    mu, logvar = _f32(mu), _f32(logvar)
    scratch = torch.empty(1024, dtype=torch.float32, device=mu.device)
    out = torch.empty((), dtype=torch.float32, device=mu.device)
    L.check(L.lib.mopoe_kl_divergence(
        L.ptr(mu), L.ptr(logvar), mu.numel(),
        0.0 if norm_value is None else float(norm_value), L.ptr(scratch), L.ptr(out),
        L.stream_ptr()), "mopoe_kl_divergence")
    return out


def reparameterize(mu, logvar, eps=None, seed=0, stream_id=0):
    mu, logvar = _f32(mu), _f32(logvar)
    eps = None if eps is None else _f32(eps)
    out = torch.empty_like(mu)
    L.check(L.lib.mopoe_reparameterize(L.ptr(mu), L.ptr(logvar), L.ptr(eps), mu.numel(),
                                       int(seed) & (2 ** 64 - 1), int(stream_id),
                                       L.ptr(out), L.stream_ptr()), "mopoe_reparameterize")
    return out


def mixture_select(mus, logvars, bounds):
    """Rows [bounds[k], bounds[k+1]) of component k (utils/utils.py:63-85)."""
    mus, logvars = _f32(mus), _f32(logvars)
    k, n, d = mus.shape
    b = torch.tensor(list(bounds), dtype=torch.int32, device=mus.device)
    out_mu = torch.empty(n, d, dtype=torch.float32, device=mus.device)
    out_lv = torch.empty_like(out_mu)
    L.check(L.lib.mopoe_mixture_select(L.ptr(mus), L.ptr(logvars), k, n, d, L.ptr(b),
                                       L.ptr(out_mu), L.ptr(out_lv), L.stream_ptr()),
            "mopoe_mixture_select")
    return out_mu, out_lv
