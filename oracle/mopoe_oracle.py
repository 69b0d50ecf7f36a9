"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference MoPoE-VAE step.

This is the *oracle* for the hot path named in BASELINE.json: a PyTorch-CPU
float32, op-for-op restatement of the reference algorithm, written from the
reference's behaviour (each function cites the reference file:line it
follows; paths are relative to /root/reference/).  It is the checker for the
HIP path -- only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import it.  The product package never does, and has no CPU fallback.

Parity pinning: the reference ships no tests or golden vectors for this path
(SURVEY.md section 4), so the oracle is pinned against outputs of the
reference itself, run in the development container by oracle/make_golden.py
and committed under tests/golden/ (tests/test_oracle_golden.py).

All arithmetic is float32 (the reference casts float64 inputs at
experiments/run_epochs.py:86).  Gradients come from torch autograd on this
restatement; Adam is restated by hand (torch.optim.Adam semantics).
"""
import math
from collections import OrderedDict
from itertools import chain, combinations

import numpy as np
import torch

HIDDEN = 256  # experiments/multimodal_cohort/networks/networks.py:14,50


# --------------------------------------------------------------------------
# configuration
# --------------------------------------------------------------------------
class Config:
    """The subset of workflow.train_exp flags the hot path reads
    (experiments/workflow.py:41-49,98-145)."""

    def __init__(self, names, input_dim, style_dim, class_dim=20,
                 method="joint_elbo", factorized=True, beta=1.0,
                 beta_style=1.0, beta_content=1.0, initial_out_logvar=-3.0,
                 learn_output_scale=True, lr=0.002, betas=(0.9, 0.999),
                 adam_eps=1e-8, poe_unimodal_elbos=True, likelihood="normal",
                 enc_layers=1, dec_layers=0, dropout=0.0, sample_scale=False,
                 gemm_operands="f32"):
        assert method in ("joint_elbo", "poe", "moe")
        # experiments/workflow.py:41-49: num_hidden_layer_encoder / _decoder, dropout_rate,
        # out_scale_per_subject (-> flags.learn_output_sample_scale, workflow.py:112)
        self.enc_layers = int(enc_layers)
        self.dec_layers = int(dec_layers)
        self.dropout = float(dropout)
        self.sample_scale = bool(sample_scale)
        # NOT a reference switch: the opt-in of the HIP path (mopoe_step.gemm_operands) whose
        # definition this restates so that it can be checked and its cost measured -- the
        # first encoder layer multiplies the bfloat16 roundings of x and W (float32 sums)
        assert gemm_operands in ("f32", "bf16")
        self.gemm_operands = gemm_operands
        # experiments/modalities/modality.py:18-30: the decoder's (loc, scale) pair feeds
        # torch.distributions.Normal or Laplace (Bernoulli / OneHotCategorical take other
        # arguments than the pair this decoder returns)
        assert likelihood in ("normal", "laplace")
        self.likelihood = likelihood
        self.names = list(names)
        self.input_dim = list(input_dim)
        M = len(self.names)
        # experiments/multimodal_cohort/experiment.py:133-136
        if isinstance(style_dim, int):
            style_dim = [style_dim] * M
        elif len(style_dim) != M:
            style_dim = [style_dim[0]] * M
        # experiments/workflow.py:148-149
        self.style_dim = list(style_dim) if factorized else [0] * M
        self.class_dim = class_dim
        self.method = method
        self.factorized = factorized
        self.beta = beta
        self.beta_style = beta_style
        self.beta_content = beta_content
        self.initial_out_logvar = initial_out_logvar
        self.learn_output_scale = learn_output_scale
        self.lr = lr
        self.betas = betas
        self.adam_eps = adam_eps
        # experiments/run_epochs.py:115: method poe adds the unimodal ELBOs (and
        # their extra forwards) only when this flag is set
        self.poe_unimodal_elbos = poe_unimodal_elbos
        # float32 is the reference's arithmetic; tests also run the oracle in
        # float64 to measure how much float32 rounding alone moves a result
        self.dtype = torch.float32

    @property
    def num_mods(self):
        return len(self.names)

    def has_style(self, m):
        return self.factorized and self.style_dim[m] > 0


def set_subsets(names):
    """experiments/utils/BaseExperiment.py:58-79: ordered powerset; keys are
    '_'.join(sorted(names)); '' (the empty set) comes first."""
    subsets = OrderedDict()
    for mod_names in chain.from_iterable(
            combinations(list(names), n) for n in range(len(names) + 1)):
        subsets["_".join(sorted(mod_names))] = sorted(mod_names)
    return subsets


# --------------------------------------------------------------------------
# parameters (state_dict naming of the reference model)
# --------------------------------------------------------------------------
def param_shapes(cfg):
    """name -> shape, in the reference's state_dict naming
    (networks.py:9-28,44-64; BaseMMVae.py:24-34)."""
    shapes = OrderedDict()
    D = cfg.class_dim
    for m, name in enumerate(cfg.names):
        d, s = cfg.input_dim[m], cfg.style_dim[m]
        e = "encoders.%s." % name
        width = d
        # nn.Sequential of (Linear, ReLU, Dropout) triples: layer l is module 3 l
        for l in range(cfg.enc_layers):
            shapes[e + "shared_encoder.%d.weight" % (3 * l)] = (HIDDEN, width)
            shapes[e + "shared_encoder.%d.bias" % (3 * l)] = (HIDDEN,)
            width = HIDDEN
        shapes[e + "class_mu.weight"] = (D, width)
        shapes[e + "class_mu.bias"] = (D,)
        shapes[e + "class_logvar.weight"] = (D, width)
        shapes[e + "class_logvar.bias"] = (D,)
        if cfg.has_style(m):
            shapes[e + "style_mu.weight"] = (s, width)
            shapes[e + "style_mu.bias"] = (s,)
            shapes[e + "style_logvar.weight"] = (s, width)
            shapes[e + "style_logvar.bias"] = (s,)
    for m, name in enumerate(cfg.names):
        d, s = cfg.input_dim[m], cfg.style_dim[m]
        k = "decoders.%s." % name
        # (a module's own parameters come before its sub-modules' in state_dict order)
        if not cfg.sample_scale:
            shapes[k + "logvar"] = (1, d)
        width = s + D
        for l in range(cfg.dec_layers):
            shapes[k + "shared_decoder.%d.weight" % (3 * l)] = (HIDDEN, width)
            shapes[k + "shared_decoder.%d.bias" % (3 * l)] = (HIDDEN,)
            width = HIDDEN
        shapes[k + "out_mu.weight"] = (d, width)
        shapes[k + "out_mu.bias"] = (d,)
        if cfg.sample_scale:        # networks.py:58-59: a Linear head instead of the parameter
            shapes[k + "logvar.weight"] = (d, width)
            shapes[k + "logvar.bias"] = (d,)
    return shapes


def init_params(cfg, seed):
    """Deterministic init that needs neither the reference nor torch's RNG:
    uniform(-1/sqrt(fan_in), 1/sqrt(fan_in)) from numpy PCG64 (the bound
    nn.Linear's default init uses), decoder logvar = initial_out_logvar.
    The golden script loads these values INTO the reference model, so the
    exact init law is immaterial for parity."""
    rng = np.random.Generator(np.random.PCG64(seed))
    params = OrderedDict()
    for name, shape in param_shapes(cfg).items():
        if name.endswith(".logvar"):
            v = np.full(shape, cfg.initial_out_logvar, dtype=np.float32)
        else:
            if name.endswith(".weight"):
                fan_in = shape[1]
            else:  # bias: fan-in of its layer
                fan_in = param_shapes(cfg)[name[:-4] + "weight"][1]
            bound = 1.0 / math.sqrt(fan_in)
            v = ((rng.random(shape) * 2.0 - 1.0) * bound).astype(np.float32)
        params[name] = torch.from_numpy(v)
    return params


def noise_rng(seed):
    """numpy PCG64 stream used for eps in fixtures and tests (float64
    standard_normal cast to float32; independent of torch's generators)."""
    return np.random.Generator(np.random.PCG64(seed))


def make_inputs(names, input_dim, N, seed, present=None):
    """Synthetic N(0,1) float32 batch {name: (N, d)}; every modality's values
    are drawn (in order) even when it is then left out of the batch."""
    rng = np.random.Generator(np.random.PCG64(seed))
    present = list(names) if present is None else present
    x = OrderedDict()
    for name, d in zip(names, input_dim):
        v = torch.from_numpy(rng.standard_normal((N, d)).astype(np.float32))
        if name in present:
            x[name] = v
    return x


def digest_stride(numel):
    """Stride of the sampled values the fixtures keep per tensor (<= 64)."""
    return max(1, -(-numel // 64))


def trainable(cfg, name):
    return cfg.learn_output_scale or not name.endswith(".logvar")


# --------------------------------------------------------------------------
# L0 math (divergence_measures/, utils/utils.py)
# --------------------------------------------------------------------------
def poe(mu, logvar, eps=1e-8):
    """experiments/divergence_measures/mm_div.py:13-20."""
    var = torch.exp(logvar) + eps
    T = 1. / var
    pd_mu = torch.sum(mu * T, dim=0) / torch.sum(T, dim=0)
    pd_var = 1. / torch.sum(T, dim=0)
    pd_logvar = torch.log(pd_var)
    return pd_mu, pd_logvar


def calc_kl_divergence(mu0, logvar0, norm_value=None):
    """experiments/divergence_measures/kl_div.py:7-14 (static-prior branch)."""
    KLD = -0.5 * torch.sum(1 - logvar0.exp() - mu0.pow(2) + logvar0)
    if norm_value is not None:
        KLD = KLD / float(norm_value)
    return KLD


def reweight_weights(w):
    """experiments/utils/utils.py:58-60."""
    return w / w.sum()


def mixture_bounds(num_samples, w):
    """Slice bounds of experiments/utils/utils.py:63-85: float32 tensor
    arithmetic int(floor(N * w_k)); the last component runs to N."""
    K = w.shape[0]
    idx_start, idx_end = [], []
    for k in range(K):
        i_start = 0 if k == 0 else int(idx_end[k - 1])
        if k == K - 1:
            i_end = num_samples
        else:
            i_end = i_start + int(torch.floor(num_samples * w[k]))
        idx_start.append(i_start)
        idx_end.append(i_end)
    idx_end[-1] = num_samples
    return idx_start, idx_end


def mixture_component_selection(mus, logvars, w):
    """experiments/utils/utils.py:63-85."""
    s, e = mixture_bounds(mus.shape[1], w)
    K = w.shape[0]
    mu_sel = torch.cat([mus[k, s[k]:e[k], :] for k in range(K)])
    logvar_sel = torch.cat([logvars[k, s[k]:e[k], :] for k in range(K)])
    return mu_sel, logvar_sel


def calc_group_divergence_moe(mus, logvars, weights, normalization):
    """experiments/divergence_measures/mm_div.py:92-111."""
    K = mus.shape[0]
    klds = torch.zeros(K, dtype=mus.dtype)
    for k in range(K):
        klds[k] = calc_kl_divergence(mus[k], logvars[k],
                                     norm_value=normalization)
    group_div = (weights * klds).sum(dim=0)
    return group_div, klds


# --------------------------------------------------------------------------
# L1 model (networks.py, BaseMMVae.py)
# --------------------------------------------------------------------------
class _Bf16Linear(torch.autograd.Function):
    """y = round_bf16(x) round_bf16(W)^T + b with float32 accumulation; the backward is the
    float32 layer's (dW = g^T x with the UNROUNDED x, db = sum g): rounding is looked
    through, as the HIP path's weight-gradient kernel does (it never sees the roundings)."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return torch.nn.functional.linear(x.bfloat16().float(), w.bfloat16().float(), b)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        return g @ w, g.t() @ x, g.sum(0)


def _hidden_stack(params, prefix, layers, h, cfg, noise, train, first_bf16=False):
    """nn.Sequential of `layers` x (Linear, ReLU, Dropout(cfg.dropout))
    (networks.py:16-20,51-55).  Dropout in training mode is ATen's
    x * (bernoulli(1 - p) / (1 - p)); the keep masks come off the noise object's
    mask tape in call order (the reference draws them from torch's global RNG)."""
    lin = torch.nn.functional.linear
    for l in range(layers):
        op = _Bf16Linear.apply if (first_bf16 and l == 0) else lin
        h = torch.relu(op(h, params[prefix + "%d.weight" % (3 * l)],
                          params[prefix + "%d.bias" % (3 * l)]))
        if train and cfg.dropout > 0.0:
            keep = noise.keep_mask(h.shape, cfg.dropout)
            h = h * (keep / (1.0 - cfg.dropout))
    return h


def encoder_forward(params, cfg, m, x, noise=None, train=False):
    """experiments/multimodal_cohort/networks/networks.py:30-36: returns
    (style_mu, style_logvar, class_mu, class_logvar, h); style is (None, None)
    when not factorized / style_dim 0; h = what the heads read."""
    e = "encoders.%s." % cfg.names[m]
    lin = torch.nn.functional.linear
    h = _hidden_stack(params, e + "shared_encoder.", cfg.enc_layers, x, cfg, noise, train,
                      first_bf16=getattr(cfg, "gemm_operands", "f32") == "bf16")
    c_mu = lin(h, params[e + "class_mu.weight"], params[e + "class_mu.bias"])
    c_lv = lin(h, params[e + "class_logvar.weight"],
               params[e + "class_logvar.bias"])
    if cfg.has_style(m):
        s_mu = lin(h, params[e + "style_mu.weight"],
                   params[e + "style_mu.bias"])
        s_lv = lin(h, params[e + "style_logvar.weight"],
                   params[e + "style_logvar.bias"])
        return s_mu, s_lv, c_mu, c_lv, h
    return None, None, c_mu, c_lv, h


def decoder_forward(params, cfg, m, z_style, z_class, noise=None, train=False):
    """experiments/multimodal_cohort/networks/networks.py:66-77: (loc, scale) with
    scale = exp(0.5*logvar): the (1, d) parameter, or (N, d) from the Linear head of
    learn_output_sample_scale."""
    k = "decoders.%s." % cfg.names[m]
    if cfg.has_style(m):
        z = torch.cat((z_style, z_class), dim=1)
    else:
        z = z_class
    h = _hidden_stack(params, k + "shared_decoder.", cfg.dec_layers, z, cfg, noise, train)
    x_hat = torch.nn.functional.linear(h, params[k + "out_mu.weight"],
                                       params[k + "out_mu.bias"])
    if cfg.sample_scale:
        logvar = torch.nn.functional.linear(h, params[k + "logvar.weight"],
                                            params[k + "logvar.bias"])
    else:
        logvar = params[k + "logvar"]
    return x_hat, (logvar * 0.5).exp()


class Noise:
    """eps source consumed in the reference's draw order
    (BaseMMVae.py:37-40,143,155-159): content first, then style per present
    modality in `modalities` order; one such group per forward call."""

    def __init__(self, tape=None, generator=None, mask_tape=None, mask_generator=None):
        self.tape = [] if tape is None else list(tape)
        self.replay = tape is not None
        self.pos = 0
        self.gen = generator
        self.dtype = torch.float32
        # dropout keep masks (0 / 1), in the order the Dropout modules run
        self.mask_tape = [] if mask_tape is None else list(mask_tape)
        self.mask_replay = mask_tape is not None
        self.mask_pos = 0
        self.mask_gen = mask_generator

    def keep_mask(self, shape, p):
        if self.mask_replay:
            keep = self.mask_tape[self.mask_pos]
            assert tuple(keep.shape) == tuple(shape), (keep.shape, shape)
            self.mask_pos += 1
        else:
            keep = torch.from_numpy(
                (self.mask_gen.random(tuple(shape)) >= p).astype(np.float32))
            self.mask_tape.append(keep)
        return keep.to(self.dtype)

    def draw(self, shape):
        if self.replay:
            eps = self.tape[self.pos]
            assert tuple(eps.shape) == tuple(shape), (eps.shape, shape)
            self.pos += 1
        else:
            eps = torch.from_numpy(self.gen.standard_normal(
                tuple(shape)).astype(np.float32))
            self.tape.append(eps)
        return eps.to(self.dtype)


def reparameterize(mu, logvar, noise):
    """experiments/utils/BaseMMVae.py:37-40."""
    std = logvar.mul(0.5).exp()
    eps = noise.draw(mu.shape)
    return eps.mul(std).add(mu)


def _modality_fusion(cfg, mus, logvars):
    """modality_fusion chosen by BaseMMVae.set_fusion_functions
    (BaseMMVae.py:43-61): poe_fusion for joint_elbo / poe (:109-122),
    moe_fusion for moe (:96-106)."""
    E, N, D = mus.shape
    if cfg.method in ("joint_elbo", "poe"):
        if cfg.method == "poe" or E == cfg.num_mods:
            mus = torch.cat((mus, torch.zeros(1, N, D, dtype=mus.dtype)), dim=0)
            logvars = torch.cat((logvars, torch.zeros(1, N, D, dtype=mus.dtype)),
                                dim=0)
        return poe(mus, logvars)
    w = reweight_weights((1 / float(E)) * torch.ones(E))
    return mixture_component_selection(mus, logvars, w)


def _fusion_condition(cfg, subset, batch):
    """BaseMMVae.py:125-134."""
    if cfg.method == "moe":
        return len(subset) == 1
    if cfg.method == "poe":
        return len(subset) == len(batch)
    return True


def inference(params, cfg, batch, sample=True, use_expert=None, noise=None, train=False):
    """experiments/utils/BaseMMVae.py:181-239 (+ encode :167-178)."""
    enc_mods = OrderedDict()
    hidden = OrderedDict()
    for m, name in enumerate(cfg.names):
        if name in batch:
            s_mu, s_lv, c_mu, c_lv, h = encoder_forward(params, cfg, m, batch[name],
                                                        noise, train)
            enc_mods[name + "_style"] = [s_mu, s_lv]
            enc_mods[name] = [c_mu, c_lv]
            hidden[name] = h
        else:
            enc_mods[name + "_style"] = [None, None]
            enc_mods[name] = [None, None]
    mus, logvars = [], []
    distr_subsets = OrderedDict()
    for s_key, mods in set_subsets(cfg.names).items():
        if s_key == "":
            continue
        if not all(name in batch for name in mods):
            continue
        mus_subset = torch.stack([enc_mods[name][0] for name in mods])
        logvars_subset = torch.stack([enc_mods[name][1] for name in mods])
        s_mu, s_logvar = _modality_fusion(cfg, mus_subset, logvars_subset)
        distr_subsets[s_key] = [s_mu, s_logvar]
        if _fusion_condition(cfg, mods, batch):
            mus.append(s_mu)
            logvars.append(s_logvar)
    mus = torch.stack(mus)
    logvars = torch.stack(logvars)
    K = mus.shape[0]
    weights = (1 / float(K)) * torch.ones(K)   # float32 as in the reference
    if sample and use_expert is None:
        joint_mu, joint_logvar = mixture_component_selection(
            mus, logvars, reweight_weights(weights))
    elif use_expert is None:
        joint_mu, joint_logvar = mus.mean(0), logvars.mean(0)
    else:
        joint_mu, joint_logvar = distr_subsets[use_expert]
    return {"modalities": enc_mods, "mus": mus, "logvars": logvars,
            "weights": weights, "joint": [joint_mu, joint_logvar],
            "subsets": distr_subsets, "_hidden": hidden}


def forward(params, cfg, batch, noise, sample_latents=True, use_expert=None, train=False):
    """experiments/utils/BaseMMVae.py:137-165.  rec[m] is (loc, scale).  `train`:
    model.train() -- the Dropout modules are live (run_epochs.py:147)."""
    latents = inference(params, cfg, batch, sample=sample_latents,
                        use_expert=use_expert, noise=noise, train=train)
    results = {"latents": latents, "group_distr": latents["joint"]}
    if sample_latents:
        class_embeddings = reparameterize(latents["joint"][0],
                                          latents["joint"][1], noise)
    else:
        class_embeddings = latents["joint"][0]
    N = latents["mus"].shape[1]
    w = reweight_weights(latents["weights"].clone())
    group_div, klds = calc_group_divergence_moe(
        latents["mus"], latents["logvars"], w, normalization=N)
    results["joint_divergence"] = group_div
    results["individual_divs"] = klds
    results["dyn_prior"] = None
    rec = OrderedDict()
    zs = OrderedDict()
    for m, name in enumerate(cfg.names):
        if name in batch:
            s_mu, s_lv = latents["modalities"][name + "_style"]
            if cfg.has_style(m) and sample_latents:
                z_s = reparameterize(s_mu, s_lv, noise)
            else:
                z_s = s_mu
            rec[name] = decoder_forward(params, cfg, m, z_s, class_embeddings, noise, train)
            zs[name] = z_s
    results["rec"] = rec
    results["_z_class"] = class_embeddings
    results["_z_style"] = zs
    return results


def normal_log_prob(loc, scale, x):
    """torch.distributions.Normal.log_prob restated
    (experiments/modalities/modality.py:42-45 calls it)."""
    var = scale ** 2
    return -((x - loc) ** 2) / (2 * var) - scale.log() \
        - math.log(math.sqrt(2 * math.pi))


def laplace_log_prob(loc, scale, x):
    """torch.distributions.Laplace.log_prob restated."""
    return -torch.log(2 * scale) - torch.abs(x - loc) / scale


def calc_log_prob(loc, scale, target, norm_value, likelihood="normal"):
    """experiments/modalities/modality.py:42-45."""
    lp = laplace_log_prob if likelihood == "laplace" else normal_log_prob
    return lp(loc, scale, target).sum() / norm_value


def calc_elbo(cfg, modality, recs, klds, present):
    """experiments/utils/utils.py:88-112 (rec/style weights of
    experiment.py:281-290: rec 1.0, style beta_style)."""
    kld_content = klds["content"]
    if modality == "joint":
        w_style_kld = 0.0
        w_rec = 0.0
        for name in cfg.names:
            if name in klds["style"]:
                w_style_kld = w_style_kld + cfg.beta_style * klds["style"][name]
                w_rec = w_rec + 1.0 * recs[name]
        kld_style, rec_error = w_style_kld, w_rec
    else:
        kld_style = cfg.beta_style * klds["style"][modality]
        rec_error = 1.0 * recs[modality]
    div = cfg.beta_content * kld_content + cfg.beta_style * kld_style
    return rec_error + cfg.beta * div


def basic_routine_epoch(params, cfg, batch, noise, train=False):
    """experiments/run_epochs.py:73-135 (+ calc_log_probs :27-38, calc_klds
    :41-48, calc_klds_style :51-59, calc_style_kld :62-69)."""
    batch = OrderedDict((k, v.to(cfg.dtype)) for k, v in batch.items())
    results = forward(params, cfg, batch, noise, train=train)
    log_probs = OrderedDict()
    weighted_log_prob = 0.0
    for name in cfg.names:
        if name in batch:
            loc, scale = results["rec"][name]
            log_probs[name] = -calc_log_prob(loc, scale, batch[name],
                                             len(batch[name]), cfg.likelihood)
            weighted_log_prob = weighted_log_prob + 1.0 * log_probs[name]
    group_divergence = results["joint_divergence"]
    klds = OrderedDict()
    for key, (mu, logvar) in results["latents"]["subsets"].items():
        klds[key] = calc_kl_divergence(mu, logvar, norm_value=len(mu))
    klds_style = OrderedDict()
    if cfg.factorized:
        for key, (mu, logvar) in results["latents"]["modalities"].items():
            if key.endswith("style") and mu is not None:
                klds_style[key] = calc_kl_divergence(mu, logvar,
                                                     norm_value=len(mu))
    if cfg.method in ("joint_elbo", "moe"):
        kld_style = 0.0
        if cfg.factorized:
            for name in cfg.names:
                if name + "_style" in klds_style:
                    kld_style = kld_style + cfg.beta_style * \
                        klds_style[name + "_style"]
        kld_weighted = cfg.beta_style * kld_style + \
            cfg.beta_content * group_divergence
        total_loss = 1.0 * weighted_log_prob + cfg.beta * kld_weighted
    else:  # poe: joint ELBO + unimodal ELBOs from extra forwards
        klds_joint = {"content": group_divergence, "style": dict()}
        elbos = OrderedDict()
        unimodal = OrderedDict()
        for name in batch.keys():
            if cfg.factorized:
                # the reference indexes klds_style[name + '_style'] and so
                # requires every present modality to have a style branch
                kld_style_m = klds_style[name + "_style"]
            else:
                kld_style_m = 0.0
            klds_joint["style"][name] = kld_style_m
            if not cfg.poe_unimodal_elbos:      # run_epochs.py:115
                continue
            r_mod = forward(params, cfg, {name: batch[name]}, noise, train=train)
            loc, scale = r_mod["rec"][name]
            log_prob_mod = -calc_log_prob(loc, scale, batch[name],
                                          len(batch[name]), cfg.likelihood)
            klds_mod = {"content": klds[name], "style": {name: kld_style_m}}
            elbos[name] = calc_elbo(cfg, name, {name: log_prob_mod}, klds_mod,
                                    batch)
            unimodal[name] = {"log_prob": log_prob_mod, "rec": (loc, scale)}
        elbos["joint"] = calc_elbo(cfg, "joint", log_probs, klds_joint, batch)
        total_loss = sum(elbos.values())
        results["_unimodal"] = unimodal
    return {"results": results, "log_probs": log_probs,
            "total_loss": total_loss, "klds": klds,
            "klds_style": klds_style}


# --------------------------------------------------------------------------
# optimiser (torch.optim.Adam semantics; experiment.py:256-279)
# --------------------------------------------------------------------------
def adam_init(params):
    """torch.optim.Adam keeps state per parameter, `step` included: a parameter
    whose .grad is None in a step (its modality was not in the batch) is
    skipped and its step count does not advance."""
    return {"step": OrderedDict((k, 0) for k in params),
            "exp_avg": OrderedDict((k, torch.zeros_like(v))
                                   for k, v in params.items()),
            "exp_avg_sq": OrderedDict((k, torch.zeros_like(v))
                                      for k, v in params.items())}


def adam_step(cfg, params, grads, state):
    """In-place torch.optim.Adam (amsgrad False, weight_decay 0,
    maximize False): p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps),
    t = the PARAMETER's own step count (torch keeps state['step'] per
    parameter and skips parameters without a gradient)."""
    b1, b2 = cfg.betas
    if not isinstance(state["step"], dict):     # one count for all: same t everywhere
        state["step"] = OrderedDict((k, state["step"]) for k in params)
    for name, p in params.items():
        g = grads.get(name)
        if g is None:
            continue
        state["step"][name] += 1
        t = state["step"][name]
        bc1 = 1 - b1 ** t
        bc2 = 1 - b2 ** t
        step_size = cfg.lr / bc1
        bc2_sqrt = math.sqrt(bc2)
        m = state["exp_avg"][name]
        v = state["exp_avg_sq"][name]
        m.lerp_(g, 1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v.sqrt() / bc2_sqrt).add_(cfg.adam_eps)
        p.addcdiv_(m, denom, value=-step_size)


def loss_and_grads(params, cfg, batch, noise):
    """One forward+backward through the restated step.  Parameters that did
    not take part (absent modality) get no entry in `grads` -- exactly the
    params whose .grad torch leaves as None, which Adam then skips."""
    leaves = OrderedDict()
    noise.dtype = cfg.dtype
    for k, v in params.items():
        leaves[k] = v.detach().to(cfg.dtype).clone().requires_grad_(
            trainable(cfg, k))
    out = basic_routine_epoch(leaves, cfg, batch, noise, train=True)
    out["total_loss"].backward()
    grads = OrderedDict((k, v.grad) for k, v in leaves.items()
                        if v.grad is not None)
    return out, grads


def train_step(params, cfg, batch, noise, state):
    out, grads = loss_and_grads(params, cfg, batch, noise)
    adam_step(cfg, params, grads, state)
    return out, grads
