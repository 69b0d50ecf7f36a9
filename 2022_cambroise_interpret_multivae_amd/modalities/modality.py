"""Mirror of the reference's modalities/modality.py."""
import os

import torch
import torch.distributions as dist


class Modality:
    """Descriptor of one data block: its encoder / decoder classes, latent
    sizes and likelihood (reference modalities/modality.py:7-52)."""

    def __init__(self, name, enc, dec, class_dim, style_dim, lhood_name):
        self.name = name
        self.encoder = enc
        self.decoder = dec
        self.class_dim = class_dim
        self.style_dim = style_dim
        self.likelihood_name = lhood_name
        self.likelihood = self.get_likelihood(lhood_name)

    def get_likelihood(self, name):
        table = {"laplace": dist.Laplace, "bernoulli": dist.Bernoulli,
                 "normal": dist.Normal, "categorical": dist.OneHotCategorical}
        if name not in table:
            raise ValueError("likelihood %r not implemented" % (name,))
        return table[name]

    def calc_log_prob(self, out_dist, target, norm_value):
        """reference modality.py:42-45"""
        return out_dist.log_prob(target).sum() / norm_value

    def save_networks(self, dir_checkpoints):
        torch.save(self.encoder.state_dict(),
                   os.path.join(dir_checkpoints, "enc_" + self.name))
        torch.save(self.decoder.state_dict(),
                   os.path.join(dir_checkpoints, "dec_" + self.name))

    def save_data(self, d, fn, args):
        raise NotImplementedError

    def plot_data(self, d):
        return d
