"""The experiment container the training loop reads (reference
multimodal_cohort/experiment.py:64-313, utils/BaseExperiment.py:6-93) -- the part on
the hot path: modalities, the ordered powerset of subsets, model(s), Adam, the loss
weights, and the reload of a trained run (`get_experiment`, experiment.py:93-121).

Out of scope: fetching / splitting the cohort
files, residualizers.  Datasets are handed in ready-made (`dataset_train` /
`dataset_test`: a ResidentCohort, a torch Dataset, or an iterable of batches);
`fit_scalers` of dataset.py reproduces the StandardScaler the reference fits."""
from collections import OrderedDict

import torch

from .. import checkpoint
from ..modalities.modality import Modality
from ..modalities.multimodal_cohort import Clinical, Rois
from ..optim import FusedAdam
from ..utils.BaseExperiment import BaseExperiment
from .dataset import MissingModalitySampler
from .networks.networks import Decoder, Encoder
from .networks.VAE import VAE


class MultimodalExperiment(BaseExperiment):
    def __init__(self, flags, dataset_train=None, dataset_test=None, mod_names=None):
        self.flags = flags
        for name, default in (("num_models", 1), ("grad_scaling", False),
                              ("poe_unimodal_elbos", True), ("start_epoch", 0)):
            if name not in vars(flags):
                setattr(flags, name, default)
        self.num_modalities = flags.num_mods
        self.modalities, self.mod_names = self.set_modalities(mod_names)
        self.subsets = self.set_subsets()
        self.dataset_train = dataset_train
        self.dataset_test = dataset_test
        self.batch_sampler_cls = MissingModalitySampler
        self.models = self.set_models()
        self.optimizers = None
        self.grad_scalers = None
        self.rec_weights = {m: 1.0 for m in self.modalities}                  # :281-286
        self.style_weights = {m: flags.beta_style for m in self.modalities}   # :288-290

    def set_modalities(self, mod_names=None):
        """experiment.py:132-144: one style dim per modality (a shorter list is its
        first entry repeated); `clinical` and `rois` for the cohort's two blocks,
        named generic blocks beyond that (the reference's own class list stops at two)."""
        f, M = self.flags, self.num_modalities
        if isinstance(f.style_dim, int):
            f.style_dim = [f.style_dim] * M
        elif len(f.style_dim) != M:
            f.style_dim = [f.style_dim[0]] * M
        mods = OrderedDict()
        cohort = (Clinical, Rois)
        for m in range(M):
            args = (Encoder, Decoder, f.class_dim, f.style_dim[m], f.likelihood)
            if mod_names is None and m < len(cohort):
                mod = cohort[m](f.input_dim[m], *args)
            else:
                mod = Modality(mod_names[m] if mod_names else "block%d" % m, *args)
            mods[mod.name] = mod
        return mods, list(mods)

    def set_models(self):
        models = [VAE(self.flags, self.modalities, self.subsets).to(self.flags.device)
                  for _ in range(self.flags.num_models)]
        return models[0] if self.flags.num_models == 1 else models

    def set_optimizers(self):
        """experiment.py:256-279: Adam(lr, betas) per model (no optimiser state is ever
        checkpointed by the reference)."""
        models = self.models if self.flags.num_models > 1 else [self.models]
        opts = [FusedAdam(m, lr=self.flags.initial_learning_rate,
                          betas=(self.flags.beta_1, self.flags.beta_2)) for m in models]
        self.optimizers = opts[0] if self.flags.num_models == 1 else opts
        self.grad_scalers = None

    @classmethod
    def get_experiment(cls, flags_file, checkpoints_dir, load_epoch=None, **datasets):
        """Rebuild a trained run: its flags file, then for every model the checkpoint
        the reference would pick (latest epoch, or by `load_epoch`)."""
        flags = checkpoint.load_flags(flags_file)
        if "num_models" not in vars(flags):
            flags.num_models = 1
        flags.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        exp = cls(flags, **datasets)
        for i in range(flags.num_models):
            many = flags.num_models > 1
            files = checkpoint.find_checkpoints(checkpoints_dir, flags.model_save,
                                                i if many else None)
            model = exp.models[i] if many else exp.models
            model.load_state_dict(checkpoint.load_state(
                checkpoint.pick_checkpoint(files, load_epoch), flags.device))
        return exp, flags
