"""Mirror of the reference's utils/utils.py (the three hot-path functions)."""
import torch

from .. import ops


def reweight_weights(w):
    """reference utils/utils.py:58-60"""
    return w / w.sum()


def mixture_bounds(num_samples, w_modalities):
    """Slice bounds exactly as the reference computes them: float32 tensor
    arithmetic int(floor(N * w_k)) (utils/utils.py:73-84)."""
    w = w_modalities.detach().float().cpu()
    K = w.shape[0]
    bounds = [0]
    for k in range(K):
        if k == K - 1:
            bounds.append(num_samples)
        else:
            bounds.append(bounds[-1] + int(torch.floor(num_samples * w[k])))
    bounds[-1] = num_samples
    return bounds


def mixture_component_selection(flags, mus, logvars, w_modalities=None):
    """Row n takes component k(n) by contiguous slices (utils/utils.py:63-85)."""
    if w_modalities is None:
        w_modalities = torch.Tensor(flags.alpha_modalities)
    return list(ops.mixture_select(mus, logvars,
                                   mixture_bounds(mus.shape[1], w_modalities)))


def calc_elbo(exp, modality, recs, klds):
    """ELBO of one modality or of the joint posterior from its already computed terms
    (reference utils/utils.py:88-112; the fused step folds the same weights into the
    loss coefficients of plan.py, this function serves callers that hold the terms):

        rec + beta * (beta_content * KL_content + beta_style * KL_style)

    `recs` / `klds["style"]` are keyed by modality name; for "joint" the reconstruction
    and style terms are the rec_weights- / style_weights-weighted sums over the
    modalities that have a style term, for a single modality its own (weight 1 on the
    reconstruction, style_weights[m] on the style KL)."""
    flags = exp.flags
    style = klds["style"]
    if modality == "joint":
        names = [m for m in exp.modalities if m in style]
        rec = sum(exp.rec_weights[m] * recs[m] for m in names)
        kl_style = sum(exp.style_weights[m] * style[m] for m in names)
    else:
        rec = 1.0 * recs[modality]
        kl_style = exp.style_weights[modality] * style[modality]
    return rec + flags.beta * (flags.beta_content * klds["content"] +
                               flags.beta_style * kl_style)
