"""Mirror of the reference's utils/BaseMMVae.py: the multimodal VAE with the
same constructor, method surface and state_dict keys -- backed by the HIP
engine.  The parameters of the encoder / decoder sub-modules are views of one
flat device buffer; `forward` / `inference` run the fused forward kernels
(no autograd graph: training goes through run_epochs.basic_routine_epoch,
whose total_loss carries the HIP backward)."""
import torch
import torch.nn as nn

from .. import checkpoint, ops
from ..engine import MoPoEEngine
from ..plan import ModelSpec
from ..divergence_measures.mm_div import calc_group_divergence_moe, poe
from . import utils


class BaseMMVae(nn.Module):
    def __init__(self, flags, modalities, subsets):
        super().__init__()
        self.num_modalities = len(modalities.keys())
        self.flags = flags
        self.modalities = modalities
        self.subsets = subsets
        names = list(modalities.keys())
        self.spec = ModelSpec.from_flags(flags, names)
        encoders = nn.ModuleDict()
        decoders = nn.ModuleDict()
        lhoods = dict()
        for m, m_key in enumerate(names):
            encoders[m_key] = modalities[m_key].encoder(flags, m)
            decoders[m_key] = modalities[m_key].decoder(flags, m)
            lhoods[m_key] = modalities[m_key].likelihood
        self.encoders = encoders
        self.decoders = decoders
        self.lhoods = lhoods
        self.engine = None
        self._adopt(torch.device(flags.device))
        self.set_fusion_functions()

    # ---------------------------------------------------------- flat storage
    def _adopt(self, device):
        """(Re)bind every parameter to a view of the engine's flat buffer."""
        old = {k: p.detach() for k, p in self.named_parameters()}
        engine = MoPoEEngine(self.spec, device)
        if set(old) != set(engine.views):
            raise RuntimeError("parameter names do not match the flat layout: %s"
                               % sorted(set(old) ^ set(engine.views)))
        if self.engine is not None:   # keep optimiser state across a device move
            engine.exp_avg.copy_(self.engine.exp_avg)
            engine.exp_avg_sq.copy_(self.engine.exp_avg_sq)
            engine.counters.copy_(self.engine.counters)
        for name, value in old.items():
            engine.views[name].copy_(value.reshape(engine.views[name].shape))
            mod_path, attr = name.rsplit(".", 1)
            mod = self.get_submodule(mod_path)
            req = mod._parameters[attr].requires_grad
            mod._parameters[attr] = nn.Parameter(engine.views[name], requires_grad=req)
        self.engine = engine
        # leaf the fused loss hangs its autograd node on (not a Parameter)
        self._anchor = torch.zeros(1, device=device, requires_grad=True)

    def _apply(self, fn, *args, **kwargs):
        super()._apply(fn, *args, **kwargs)
        dev = next(self.parameters()).device
        p0 = next(self.parameters())
        base = self.engine.params
        inside = (p0.device == base.device and p0.data_ptr() >= base.data_ptr() and
                  p0.data_ptr() < base.data_ptr() + base.numel() * 4)
        if not inside:
            self._adopt(dev)
        return self

    # ------------------------------------------------- reference method surface
    def reparameterize(self, mu, logvar):
        """reference BaseMMVae.py:37-40 (eps from on-device Philox)."""
        self.engine._calls += 1
        return ops.reparameterize(mu, logvar, None, seed=self.engine.seed,
                                  stream_id=self.engine._calls)

    def set_fusion_functions(self):
        """reference BaseMMVae.py:43-61"""
        weights = utils.reweight_weights(torch.Tensor(self.flags.alpha_modalities))
        self.weights = weights.to(self.flags.device)
        if self.flags.modality_moe:
            self.modality_fusion = self.moe_fusion
            self.fusion_condition = self.fusion_condition_moe
        elif self.flags.modality_poe:
            self.modality_fusion = self.poe_fusion
            self.fusion_condition = self.fusion_condition_poe
        elif self.flags.joint_elbo:
            self.modality_fusion = self.poe_fusion
            self.fusion_condition = self.fusion_condition_joint
        else:
            raise NotImplementedError("method jsd is outside the hot path")
        self.calc_joint_divergence = self.divergence_static_prior

    def divergence_static_prior(self, mus, logvars, weights=None):
        """reference BaseMMVae.py:64-78"""
        if weights is None:
            weights = self.weights
        weights = utils.reweight_weights(weights.clone())
        div_measures = calc_group_divergence_moe(self.flags, mus, logvars, weights,
                                                 normalization=mus.shape[1])
        return {"joint_divergence": div_measures[0], "individual_divs": div_measures[1],
                "dyn_prior": None}

    def moe_fusion(self, mus, logvars, weights=None):
        """reference BaseMMVae.py:96-106"""
        if weights is None:
            weights = self.weights
        weights = utils.reweight_weights(weights)
        return utils.mixture_component_selection(self.flags, mus, logvars, weights)

    def poe_fusion(self, mus, logvars, weights=None):
        """reference BaseMMVae.py:109-122"""
        if self.flags.modality_poe or mus.shape[0] == len(self.modalities.keys()):
            zeros = torch.zeros(1, mus.shape[1], self.flags.class_dim, device=mus.device)
            mus = torch.cat((mus, zeros), dim=0)
            logvars = torch.cat((logvars, zeros), dim=0)
        return list(poe(mus, logvars))

    def fusion_condition_moe(self, subset, input_batch=None):
        return len(subset) == 1

    def fusion_condition_poe(self, subset, input_batch=None):
        return len(subset) == len(input_batch.keys())

    def fusion_condition_joint(self, subset, input_batch=None):
        return True

    def forward(self, input_batch, sample_latents=True, use_expert=None):
        """reference BaseMMVae.py:137-165: returns the same results dict
        (latents, group_distr, joint_divergence, individual_divs, dyn_prior,
        rec{m: Normal}), computed by the fused HIP forward."""
        plan, ws = self.engine.forward(input_batch, sample=sample_latents,
                                       use_expert=use_expert, fresh=True)
        return self.engine.results(plan, ws)

    def encode(self, input_batch):
        """reference BaseMMVae.py:167-178"""
        return self.inference(input_batch)["modalities"]

    def inference(self, input_batch, num_samples=None, sample=True, use_expert=None):
        """reference BaseMMVae.py:181-239"""
        plan, ws = self.engine.forward(input_batch, sample=sample, use_expert=use_expert,
                                       fresh=True)
        return self.engine.results(plan, ws)["latents"]

    # ------------------------------------------------------------ generation
    # (reference BaseMMVae.py:242-322.  Everything below is "draw latents, run the
    # decoders": engine.decode is one mopoe_linear per modality on [style | content].)
    def _prior_draw(self, num_samples, dim):
        """z ~ N(0, I) of shape (num_samples, dim) from the device generator."""
        zero = torch.zeros(num_samples, dim, device=self.engine.device)
        return self.reparameterize(zero, zero)

    def get_random_style_dists(self, num_samples):
        """{modality: [mu, logvar]} of the style prior N(0, I) (reference :293-303)."""
        dev = self.engine.device
        return {name: [torch.zeros(num_samples, mod.style_dim, device=dev),
                       torch.zeros(num_samples, mod.style_dim, device=dev)]
                for name, mod in self.modalities.items()}

    def get_random_styles(self, num_samples):
        """{modality: style sample from the prior, or None without a style branch}
        (reference :306-316)."""
        return {name: self._prior_draw(num_samples, self.spec.style_dim[m])
                if self.spec.has_style(m) else None
                for m, name in enumerate(self.modalities)}

    def generate_sufficient_statistics_from_latents(self, latents):
        """{modality: likelihood(loc, scale)} for content / style latents
        (reference :258-266)."""
        decoded = self.engine.decode(latents["content"], latents["style"])
        return {name: self.lhoods[name](loc, scale, validate_args=False)
                for name, (loc, scale) in decoded.items()}

    def generate_from_latents(self, latents):
        """{modality: mean of its likelihood} (reference :269-275)."""
        stats = self.generate_sufficient_statistics_from_latents(latents)
        return {name: stats[name].mean for name in latents["style"]}

    def generate(self, num_samples=None):
        """Unconditional samples: content and styles from the prior (reference :242-255)."""
        n = self.flags.batch_size if num_samples is None else num_samples
        return self.generate_from_latents(
            {"content": self._prior_draw(n, self.flags.class_dim),
             "style": self.get_random_styles(n)})

    def cond_generation(self, latent_distributions, num_samples=None):
        """{key: generations} for every given content posterior [mu, logvar]; one draw
        of prior styles serves all keys (reference :278-290)."""
        if num_samples is None:
            num_samples = len(next(iter(latent_distributions.values()))[0])
        styles = self.get_random_styles(num_samples)
        return {key: self.generate_from_latents(
                    {"content": self.reparameterize(mu=mu, logvar=logvar), "style": styles})
                for key, (mu, logvar) in latent_distributions.items()}

    def save_networks(self):
        """Per-modality state dicts `enc_<name>` / `dec_<name>` under
        flags.dir_checkpoints (reference :315-322)."""
        checkpoint.save_networks(self, self.flags.dir_checkpoints)
