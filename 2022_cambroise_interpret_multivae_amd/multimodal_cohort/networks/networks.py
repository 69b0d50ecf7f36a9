"""Mirror of the reference's multimodal_cohort/networks/networks.py: the MLP
encoder / decoder, with the same constructor signature, sub-module names and
state_dict keys.  Standalone forward() runs the HIP linear kernel; inside a
VAE the parameters are views of the model's flat device buffer and the fused
step never calls these forwards."""
import torch
import torch.nn as nn

from ... import ops

HIDDEN = 256


def _check_topology(flags):
    if flags.num_hidden_layer_encoder != 1 or flags.num_hidden_layer_decoder != 0:
        raise NotImplementedError(
            "the HIP path implements the train_exp default topology: one hidden "
            "encoder layer, no hidden decoder layer (reference workflow.py:41-49)")
    if getattr(flags, "learn_output_sample_scale", False):
        raise NotImplementedError("learn_output_sample_scale")
    if getattr(flags, "dropout_rate", 0.0) != 0.0:
        raise NotImplementedError("dropout_rate != 0")


class Encoder(nn.Module):
    """reference networks.py:4-36"""

    def __init__(self, flags, mod_num):
        super().__init__()
        _check_topology(flags)
        self.flags = flags
        self.shared_encoder = nn.Sequential(
            nn.Linear(flags.input_dim[mod_num], HIDDEN), nn.ReLU(),
            nn.Dropout(flags.dropout_rate))
        self.style_dim = flags.style_dim[mod_num]
        self.class_mu = nn.Linear(HIDDEN, flags.class_dim)
        self.class_logvar = nn.Linear(HIDDEN, flags.class_dim)
        if flags.factorized_representation and self.style_dim > 0:
            self.style_mu = nn.Linear(HIDDEN, self.style_dim)
            self.style_logvar = nn.Linear(HIDDEN, self.style_dim)

    def forward(self, h):
        lin = self.shared_encoder[0]
        h = ops.linear(h, lin.weight, lin.bias, relu=True)
        c_mu = ops.linear(h, self.class_mu.weight, self.class_mu.bias)
        c_lv = ops.linear(h, self.class_logvar.weight, self.class_logvar.bias)
        if self.flags.factorized_representation and self.style_dim > 0:
            return (ops.linear(h, self.style_mu.weight, self.style_mu.bias),
                    ops.linear(h, self.style_logvar.weight, self.style_logvar.bias),
                    c_mu, c_lv)
        return None, None, c_mu, c_lv


class Decoder(nn.Module):
    """reference networks.py:39-77"""

    def __init__(self, flags, mod_num):
        super().__init__()
        _check_topology(flags)
        self.flags = flags
        self.shared_decoder = nn.Sequential()
        self.style_dim = flags.style_dim[mod_num]
        self.out_mu = nn.Linear(self.style_dim + flags.class_dim,
                                flags.input_dim[mod_num])
        self.logvar = nn.Parameter(
            data=torch.FloatTensor(1, flags.input_dim[mod_num]).fill_(
                flags.initial_out_logvar),
            requires_grad=flags.learn_output_scale)

    def forward(self, style_latent_space, class_latent_space):
        if self.flags.factorized_representation and self.style_dim > 0:
            z = torch.cat((style_latent_space, class_latent_space), dim=1)
        else:
            z = class_latent_space
        x_hat = ops.linear(z, self.out_mu.weight, self.out_mu.bias)
        return x_hat, (self.logvar.detach() * 0.5).exp().to(z.device)
