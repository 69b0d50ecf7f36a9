"""Mirror of the reference's multimodal_cohort/networks/networks.py: the MLP
encoder / decoder, with the same constructor signature, sub-module names and
state_dict keys.  Standalone forward() runs the HIP linear kernel; inside a
VAE the parameters are views of the model's flat device buffer and the fused
step never calls these forwards."""
import torch
import torch.nn as nn

from ... import ops

HIDDEN = 256


def _stack(input_dim, layers, dropout_rate):
    """`layers` x (Linear(., 256), ReLU, Dropout): the Linear of layer l is module 3 l
    (reference networks.py:16-20,51-55; the state_dict keys hang on these indices)."""
    seq = nn.Sequential()
    for _ in range(layers):
        seq.append(nn.Linear(input_dim, HIDDEN))
        seq.append(nn.ReLU())
        seq.append(nn.Dropout(dropout_rate))
        input_dim = HIDDEN
    return seq, input_dim


def _run_stack(seq, h, training):
    """Standalone forward of a stack on the HIP linear kernel.  Outside the fused /
    general training step (whose kernels draw or take the masks themselves) a live
    Dropout is torch's own."""
    for mod in seq:
        if isinstance(mod, nn.Linear):
            h = ops.linear(h, mod.weight, mod.bias, relu=True)
        elif isinstance(mod, nn.Dropout) and training and mod.p > 0:
            h = torch.nn.functional.dropout(h, mod.p, True)
    return h


class Encoder(nn.Module):
    """reference networks.py:4-36"""

    def __init__(self, flags, mod_num):
        super().__init__()
        self.flags = flags
        self.shared_encoder, width = _stack(flags.input_dim[mod_num],
                                            flags.num_hidden_layer_encoder, flags.dropout_rate)
        self.style_dim = flags.style_dim[mod_num]
        self.class_mu = nn.Linear(width, flags.class_dim)
        self.class_logvar = nn.Linear(width, flags.class_dim)
        if flags.factorized_representation and self.style_dim > 0:
            self.style_mu = nn.Linear(width, self.style_dim)
            self.style_logvar = nn.Linear(width, self.style_dim)

    def forward(self, h):
        h = _run_stack(self.shared_encoder, h, self.training)
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
        self.flags = flags
        self.style_dim = flags.style_dim[mod_num]
        self.shared_decoder, width = _stack(self.style_dim + flags.class_dim,
                                            flags.num_hidden_layer_decoder, flags.dropout_rate)
        self.out_mu = nn.Linear(width, flags.input_dim[mod_num])
        if getattr(flags, "learn_output_sample_scale", False):
            self.logvar = nn.Linear(width, flags.input_dim[mod_num])
        else:
            self.logvar = nn.Parameter(
                data=torch.FloatTensor(1, flags.input_dim[mod_num]).fill_(
                    flags.initial_out_logvar),
                requires_grad=flags.learn_output_scale)

    def forward(self, style_latent_space, class_latent_space):
        if self.flags.factorized_representation and self.style_dim > 0:
            z = torch.cat((style_latent_space, class_latent_space), dim=1)
        else:
            z = class_latent_space
        h = _run_stack(self.shared_decoder, z, self.training)
        x_hat = ops.linear(h, self.out_mu.weight, self.out_mu.bias)
        if getattr(self.flags, "learn_output_sample_scale", False):
            logvar = ops.linear(h, self.logvar.weight, self.logvar.bias)
        else:
            logvar = self.logvar.detach()
        return x_hat, (logvar * 0.5).exp().to(z.device)
