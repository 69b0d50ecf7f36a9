"""TEST INFRASTRUCTURE ONLY -- loader for the *reference* Python package.

This module exists only in the development container, where the upstream
reference lives at /root/reference.  It is used by `oracle/make_golden.py`
(to generate the committed fixtures under tests/golden/) and by
`tests/test_oracle_vs_reference.py` (skipped when /root/reference is absent,
as it is on the GPU box).  Nothing in the product package imports it.

The reference imports five third-party packages that are absent from this
image and carry none of the hot-path arithmetic (torchvision, tensorboardX,
iterstrat, statsmodels, fire; plus seaborn / nilearn / imageio for modules
off the path).  They are replaced by inert stub modules before the first
reference import.  No reference source or bytecode is copied or written.
"""
import importlib
import os
import sys
import types
from collections import OrderedDict
from itertools import chain, combinations

import torch

REFERENCE_ROOT = "/root/reference/experiments"


def reference_available():
    return os.path.isdir(REFERENCE_ROOT)


def _stub(name, **attrs):
    mod = sys.modules.get(name)
    if mod is None:
        mod = types.ModuleType(name)
        mod.__path__ = []  # behave like a package for dotted imports
        sys.modules[name] = mod
    for k, v in attrs.items():
        setattr(mod, k, v)
    return mod


def install_stubs():
    """Register inert stand-ins for the absent third-party imports."""
    class _Nop:
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            return None

        def __getattr__(self, name):
            return _Nop()

    class _Module(torch.nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    _stub("torchvision")
    _stub("torchvision.utils", save_image=_Nop(), make_grid=_Nop())
    _stub("torchvision.transforms", Compose=_Nop, ToTensor=_Nop)
    _stub("torchvision.models", inception_v3=_Nop())
    _stub("torchvision.models.inception", InceptionA=_Module,
          InceptionC=_Module, InceptionE=_Module)
    sys.modules["torchvision"].utils = sys.modules["torchvision.utils"]
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["torchvision"].models = sys.modules["torchvision.models"]
    sys.modules["torchvision.models"].inception = \
        sys.modules["torchvision.models.inception"]
    _stub("tensorboardX", SummaryWriter=_Nop)
    _stub("iterstrat")
    _stub("iterstrat.ml_stratifiers", MultilabelStratifiedShuffleSplit=_Nop,
          MultilabelStratifiedKFold=_Nop)
    _stub("statsmodels")
    _stub("statsmodels.api", OLS=_Nop, MixedLM=_Nop)
    _stub("statsmodels.stats")
    _stub("statsmodels.stats.anova", anova_lm=_Nop())
    _stub("fire", Fire=_Nop())
    _stub("imageio", imread=_Nop())
    _stub("seaborn")
    _stub("nilearn", plotting=_Nop(), datasets=_Nop())
    _stub("nilearn.plotting")
    _stub("nilearn.datasets")


def import_reference():
    """Returns a namespace of the reference modules on the hot path."""
    if not reference_available():
        raise RuntimeError("reference not present at %s" % REFERENCE_ROOT)
    sys.dont_write_bytecode = True
    install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    ns = types.SimpleNamespace()
    ns.kl_div = importlib.import_module("divergence_measures.kl_div")
    ns.mm_div = importlib.import_module("divergence_measures.mm_div")
    ns.utils = importlib.import_module("utils.utils")
    ns.BaseMMVae = importlib.import_module("utils.BaseMMVae")
    ns.VAE = importlib.import_module("multimodal_cohort.networks.VAE")
    ns.networks = importlib.import_module("multimodal_cohort.networks.networks")
    ns.modality = importlib.import_module("modalities.modality")
    ns.run_epochs = importlib.import_module("run_epochs")
    return ns


def make_flags(input_dim, style_dim, class_dim=20, method="joint_elbo",
               factorized=True, beta=1.0, beta_style=1.0, beta_content=1.0,
               batch_size=256, initial_out_logvar=-3.0, learn_output_scale=True,
               poe_unimodal_elbos=True, likelihood="normal", enc_layers=1, dec_layers=0,
               dropout=0.0, sample_scale=False):
    """SimpleNamespace mirroring what workflow.train_exp builds
    (reference experiments/workflow.py:98-145)."""
    M = len(input_dim)
    flags = types.SimpleNamespace(
        batch_size=batch_size, beta=beta, beta_content=beta_content,
        beta_style=beta_style, class_dim=class_dim,
        factorized_representation=factorized, input_dim=list(input_dim),
        joint_elbo=(method == "joint_elbo"), modality_jsd=False,
        modality_moe=(method == "moe"), modality_poe=(method == "poe"),
        poe_unimodal_elbos=poe_unimodal_elbos, num_hidden_layer_encoder=enc_layers,
        num_hidden_layer_decoder=dec_layers, dropout_rate=dropout,
        initial_out_logvar=initial_out_logvar,
        learn_output_scale=learn_output_scale,
        learn_output_sample_scale=sample_scale, likelihood=likelihood,
        style_dim=list(style_dim) if factorized else [0] * M,
        num_models=1, num_mods=M, device=torch.device("cpu"),
        alpha_modalities=[1.0 / (M + 1)] * (M + 1), grad_scaling=False)
    return flags


def build_experiment(ns, flags, names):
    """Bare object exposing the attributes the reference loop reads
    (SURVEY.md Appendix B step 4).  `names` are modality names in order."""
    Modality = ns.modality.Modality

    class _Generic(Modality):
        def save_data(self, d, fn, args):
            pass

        def plot_data(self, d):
            return d

    mods = OrderedDict()
    for m, name in enumerate(names):
        mods[name] = _Generic(name, ns.networks.Encoder, ns.networks.Decoder,
                              flags.class_dim, flags.style_dim[m],
                              flags.likelihood)
    exp = types.SimpleNamespace()
    exp.flags = flags
    exp.modalities = mods
    # BaseExperiment.set_subsets (reference utils/BaseExperiment.py:58-79) is a
    # method of an ABC with abstract members; call it unbound on our object.
    base_exp = importlib.import_module("utils.BaseExperiment")
    exp.subsets = base_exp.BaseExperiment.set_subsets(exp)
    exp.models = ns.VAE.VAE(flags, mods, exp.subsets)
    exp.rec_weights = {m: 1.0 for m in mods}
    exp.style_weights = {m: flags.beta_style for m in mods}
    return exp


class NoiseTape:
    """Record / replay the eps draws of BaseMMVae.reparameterize.
    `generator` is a numpy Generator (see mopoe_oracle.noise_rng)."""

    def __init__(self, model, generator=None, replay=None):
        self.tape = [] if replay is None else list(replay)
        self.replay = replay is not None
        self.pos = 0
        self.gen = generator
        model.reparameterize = self

    def __call__(self, mu, logvar):
        std = logvar.mul(0.5).exp()
        if self.replay:
            eps = self.tape[self.pos]
            self.pos += 1
        else:
            import numpy as np
            eps = torch.from_numpy(self.gen.standard_normal(
                tuple(mu.shape)).astype(np.float32))
            self.tape.append(eps)
        return eps.mul(std).add(mu)


class _TapeDropout(torch.nn.Module):
    """Stands in for one nn.Dropout of the reference model: the same arithmetic as ATen's
    dropout in training mode -- x * (bernoulli(1 - p) / (1 - p)) -- with the keep mask
    taken from a tape instead of torch's global generator."""

    def __init__(self, p, tape):
        super().__init__()
        self.p = p
        self.tape = tape

    def forward(self, x):
        if not self.training or self.p == 0.0:
            return x
        return x * (self.tape.draw(x.shape, self.p) / (1.0 - self.p))


class MaskTape:
    """Record / replay the dropout keep masks of a reference model, in the order its
    Dropout modules run (networks.py:19,54).  `generator`: a numpy Generator."""

    def __init__(self, model, generator=None, replay=None):
        self.tape = [] if replay is None else list(replay)
        self.replay = replay is not None
        self.pos = 0
        self.gen = generator
        for parent in list(model.modules()):
            for name, child in list(parent.named_children()):
                if isinstance(child, (torch.nn.Dropout, _TapeDropout)):
                    setattr(parent, name, _TapeDropout(child.p, self))

    def draw(self, shape, p):
        if self.replay:
            keep = self.tape[self.pos]
            self.pos += 1
        else:
            import numpy as np
            keep = torch.from_numpy(
                (self.gen.random(tuple(shape)) >= p).astype(np.float32))
            self.tape.append(keep)
        return keep
