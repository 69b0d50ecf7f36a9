"""Builds the reference-shaped experiment object (flags, modalities, subsets,
models, optimizers, weights) out of the package's mirror classes."""
import types
from collections import OrderedDict

import torch

import mopoe_amd as mm
from importlib import import_module

_P = "2022_cambroise_interpret_multivae_amd."
networks = import_module(_P + "multimodal_cohort.networks.networks")
VAE = import_module(_P + "multimodal_cohort.networks.VAE").VAE
modality = import_module(_P + "modalities.modality")
cohort_mods = import_module(_P + "modalities.multimodal_cohort")
base_exp = import_module(_P + "utils.BaseExperiment")
run_epochs = import_module(_P + "run_epochs")
optim = import_module(_P + "optim")


def make_flags(cfg, device):
    """The SimpleNamespace reference workflow.train_exp assembles
    (workflow.py:98-145), from an oracle Config."""
    M = cfg.num_mods
    return types.SimpleNamespace(
        batch_size=256, beta=cfg.beta, beta_content=cfg.beta_content,
        beta_style=cfg.beta_style, class_dim=cfg.class_dim,
        factorized_representation=cfg.factorized, input_dim=list(cfg.input_dim),
        joint_elbo=cfg.method == "joint_elbo", modality_jsd=False,
        modality_moe=cfg.method == "moe", modality_poe=cfg.method == "poe",
        poe_unimodal_elbos=True, num_hidden_layer_encoder=cfg.enc_layers,
        num_hidden_layer_decoder=cfg.dec_layers, dropout_rate=cfg.dropout,
        initial_out_logvar=cfg.initial_out_logvar,
        learn_output_scale=cfg.learn_output_scale,
        learn_output_sample_scale=cfg.sample_scale, likelihood="normal", style_dim=list(cfg.style_dim), num_models=1, num_mods=M,
        device=torch.device(device), alpha_modalities=[1.0 / (M + 1)] * (M + 1),
        grad_scaling=False, initial_learning_rate=cfg.lr, beta_1=cfg.betas[0],
        beta_2=cfg.betas[1], start_epoch=0, end_epoch=1, dir_checkpoints="/tmp",
        model_save="model")


def make_experiment(cfg, device):
    flags = make_flags(cfg, device)
    mods = OrderedDict()
    for m, name in enumerate(cfg.names):
        mods[name] = modality.Modality(name, networks.Encoder, networks.Decoder,
                                       flags.class_dim, flags.style_dim[m], "normal")
    exp = types.SimpleNamespace()
    exp.flags = flags
    exp.modalities = mods
    exp.subsets = base_exp.set_subsets(mods)
    exp.models = VAE(flags, mods, exp.subsets).to(flags.device)
    exp.optimizers = optim.FusedAdam(exp.models, lr=flags.initial_learning_rate,
                                     betas=(flags.beta_1, flags.beta_2))
    exp.rec_weights = {m: 1.0 for m in mods}
    exp.style_weights = {m: flags.beta_style for m in mods}
    return exp
