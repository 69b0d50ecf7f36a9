"""Mirror of the reference's divergence_measures/mm_div.py (hot-path part)."""
import torch

from .. import ops
from .kl_div import calc_kl_divergence


def poe(mu, logvar, eps=1e-8):
    """Product of Gaussian experts over axis 0 (reference mm_div.py:13-20)."""
    return ops.poe(mu, logvar, eps)


def calc_group_divergence_moe(flags, mus, logvars, weights, normalization=None):
    """sum_k w_k KL_k and the (K,) KLs (reference mm_div.py:92-111)."""
    if normalization is None:
        raise NotImplementedError("per-sample (unnormalised) group divergence is "
                                  "not used by the training path")
    klds = torch.stack([calc_kl_divergence(mus[k], logvars[k], norm_value=normalization)
                        for k in range(mus.shape[0])])
    weights = weights.to(klds.device)
    return (weights * klds).sum(dim=0), klds
