"""Checkpoint interoperability with the reference.

Files and layout are the reference's, so a model trained by either side loads into
the other:

    <run>/flags.rar                          torch.save(flags)     utils/utils.py:115-125
    <dir_checkpoints>/<epoch:04d>/<model_save>         model.state_dict()   run_epochs.py:243-256
    <dir_checkpoints>/model_<i>/<epoch:04d>/...        (num_models > 1)
    <dir_checkpoints>/enc_<name>, dec_<name>           per-modality networks  BaseMMVae.py:315-322

Everything is read with loaders that execute nothing from the file:
`torch.load(..., weights_only=True)` for tensors, and for the flags file the same
loader with `types.SimpleNamespace` allow-listed (the reference pickles the
namespace `workflow.train_exp` assembles; it holds numbers, strings, lists and a
torch.device)."""
import glob
import os
import types

import torch


def _epoch_of(path):
    return int(path.split(os.sep)[-2])


def checkpoint_dir(flags, epoch, model_idx=None):
    parts = [flags.dir_checkpoints]
    if model_idx is not None:
        parts.append("model_%d" % model_idx)
    parts.append("%04d" % epoch)
    return os.path.join(*parts)


def save_flags(flags, path):
    """The reference's flags file (torch.save of the namespace)."""
    torch.save(flags, path)


def load_flags(path):
    with torch.serialization.safe_globals([types.SimpleNamespace]):
        flags = torch.load(path, weights_only=True)
    if not isinstance(flags, types.SimpleNamespace):
        raise TypeError("%s does not hold a flags namespace" % path)
    return flags


def save_networks(model, dir_checkpoints):
    """Per-modality encoder / decoder state dicts, `enc_<name>` / `dec_<name>`."""
    for name in model.modalities:
        torch.save(model.encoders[name].state_dict(),
                   os.path.join(dir_checkpoints, "enc_" + name))
        torch.save(model.decoders[name].state_dict(),
                   os.path.join(dir_checkpoints, "dec_" + name))


def save_model(model, flags, epoch, model_idx=None):
    """State dict of the whole model under its epoch directory + the per-modality
    networks next to the epoch directories."""
    d = checkpoint_dir(flags, epoch, model_idx)
    os.makedirs(d, exist_ok=True)
    save_networks(model, flags.dir_checkpoints)
    path = os.path.join(d, flags.model_save)
    torch.save(model.state_dict(), path)
    return path


def find_checkpoints(checkpoints_dir, model_save, model_idx=None):
    """Checkpoint files of one model, oldest epoch first."""
    pattern = [checkpoints_dir]
    if model_idx is not None:
        pattern.append("model_%d" % model_idx)
    pattern += ["*", model_save]
    return sorted(glob.glob(os.path.join(*pattern)), key=_epoch_of)


def pick_checkpoint(files, load_epoch=None):
    """The file the reference's get_experiment picks (experiment.py:103-119): the
    latest one, or for `load_epoch` the entry at argmin(epochs >= load_epoch) -- the
    first file whose epoch is below load_epoch, the oldest file when there is none."""
    if not files:
        raise ValueError("You need first to train the model.")
    if load_epoch is None:
        return files[-1]
    for f in files:
        if _epoch_of(f) < load_epoch:
            return f
    return files[0]


def load_state(path, device):
    return torch.load(path, map_location=device, weights_only=True)
