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
    """reference utils/utils.py:88-112"""
    flags = exp.flags
    mods = exp.modalities
    s_weights = exp.style_weights
    r_weights = exp.rec_weights
    kld_content = klds["content"]
    if modality == "joint":
        w_style_kld = 0.0
        w_rec = 0.0
        klds_style = klds["style"]
        for m_key in mods.keys():
            if m_key in klds_style.keys():
                w_style_kld += s_weights[m_key] * klds_style[m_key]
                w_rec += r_weights[m_key] * recs[m_key]
        kld_style = w_style_kld
        rec_error = w_rec
    else:
        kld_style = s_weights[modality] * klds["style"][modality]
        rec_error = 1.0 * recs[modality]
    div = flags.beta_content * kld_content + flags.beta_style * kld_style
    return rec_error + flags.beta * div
