"""Batched inference for the digital-avatar analysis.

The reference's `daa` workflow (workflow.py:388-419) runs, per validation fold,
`M` separate stochastic forwards of one test batch (M = 1000) and then
`n_samples x n_scores` forwards (150 x 7) of copies of the batch in which one
clinical score is replaced by a sampled value -- about two thousand launches of a
50-row forward.  Here each of the two groups is ONE launch: the repeats are folded
into the batch axis (`mopoe_step.group_rows` = rows of the original batch, so the
row-position rules of `mixture_component_selection` apply per repeat exactly as
in the separate forwards), the unchanged modalities are not copied (the kernels
gather them through `row_index`), and every repeat draws its own eps.

Only `rec[m].loc` / `.scale` are needed by the workflow, and that is what these
helpers return (device tensors).
"""
from collections import OrderedDict

import torch


def _engine(model):
    eng = getattr(model, "engine", None)
    if eng is None:
        raise TypeError("expected a HIP-backed BaseMMVae (model.engine is missing)")
    return eng


def _scale(model, name, ws=None, m=None, shape=None):
    """The likelihood's scale of a folded launch: the learnt (1, d) vector, or -- with
    learn_output_sample_scale (networks.py:57-59,73-75: a logvar HEAD, so the scale is per
    sample and per forward, which the reference's loop reads as `.scale` of every forward,
    workflow.py:388-398) -- exp(logvar / 2) of the launch's own logvar-head output, viewed
    like `loc`."""
    if model.engine.spec.sample_scale:
        rows = shape[0] * shape[1]
        return (ws.lv[m][:rows] * 0.5).exp().view(*shape)
    return (model.engine.views["decoders.%s.logvar" % name] * 0.5).exp()


def repeated_reconstructions(model, data, M, sample_latents=True):
    """`M` forwards of the batch `data` as one launch (workflow.py:388-396).

    Returns {modality: (loc (M, n, d_m), scale (1, d_m))}: `loc[i]` is
    `model(data, sample_latents)["rec"][modality].loc` of the i-th forward; with
    learn_output_sample_scale `scale` is (M, n, d_m), `scale[i]` that forward's `.scale`."""
    eng = _engine(model)
    names = list(data.keys())
    n = data[names[0]].shape[0]
    idx = torch.arange(n, device=eng.device, dtype=torch.int32).repeat(M)
    plan, ws = eng.forward(data, sample=sample_latents, row_index=idx,
                           group_rows=n, fresh=True)
    out = OrderedDict()
    for m, name in enumerate(eng.spec.names):
        if name in plan.present:
            d = eng.spec.input_dim[m]
            out[name] = (ws.loc[m][:M * n].view(M, n, d), _scale(model, name, ws, m, (M, n, d)))
    return out


def mean_reconstructions(model, data, M):
    """The averages the workflow keeps of the M stochastic forwards
    (workflow.py:397-399): {modality: (mean loc (n, d_m), scale (1, d_m))}.
    (The mean over M of a scale that does not depend on the sample is the scale; a per-sample
    scale -- learn_output_sample_scale -- is averaged over the forwards as the reference does,
    workflow.py:398: (n, d_m).)"""
    rec = repeated_reconstructions(model, data, M, sample_latents=True)
    return OrderedDict((k, (loc.mean(0), scale.mean(0) if scale.dim() == 3 else scale))
                       for k, (loc, scale) in rec.items())


def perturbed_reconstructions(model, data, scores_values, sampling_strategy="likelihood",
                              sample_latents=True, modality="clinical", target="rois"):
    """The digital avatars of one fold as one launch (workflow.py:405-419).

    For every (sample_idx, score idx) the reference forwards a copy of the batch
    whose column `idx` of `data[modality]` is `scores_values[sample_idx, :, idx]`
    (strategy "likelihood": scores_values is (n_samples, n, n_scores)) or
    `scores_values[:, sample_idx, idx]` (otherwise: (n, n_samples, n_scores)) and
    keeps `rec[target].loc`.  Returns the avatars as the workflow's array layout
    (n, n_scores, n_samples, d_target), on the device."""
    eng = _engine(model)
    x = data[modality].to(eng.device, dtype=torch.float32)
    n, n_scores = x.shape
    sv = scores_values.to(eng.device, dtype=torch.float32)
    if sampling_strategy != "likelihood":
        sv = sv.permute(1, 0, 2)                      # -> (n_samples, n, n_scores)
    n_samples = sv.shape[0]
    if tuple(sv.shape) != (n_samples, n, n_scores):
        raise ValueError("scores_values has shape %s" % (tuple(scores_values.shape),))
    # (sample, score, row, column): column `score` of repeat (sample, score) replaced
    big = x.expand(n_samples, n_scores, n, n_scores).clone()
    ar = torch.arange(n_scores, device=eng.device)
    big[:, ar, :, ar] = sv.permute(2, 0, 1)           # [score, sample, row]
    R = n_samples * n_scores
    batch = OrderedDict()
    rows = OrderedDict()
    for name in data:
        if name == modality:
            batch[name] = big.view(R * n, n_scores)
            rows[name] = None
        else:   # unchanged modality: gathered, not copied
            batch[name] = data[name]
            rows[name] = torch.arange(n, device=eng.device, dtype=torch.int32).repeat(R)
    plan, ws = eng.forward(batch, sample=sample_latents, row_index=rows,
                           group_rows=n, fresh=True)
    mt = eng.spec.names.index(target)
    d = eng.spec.input_dim[mt]
    loc = ws.loc[mt][:R * n].view(n_samples, n_scores, n, d)
    return loc.permute(2, 1, 0, 3)
