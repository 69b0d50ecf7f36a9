"""Host logic of the hot path: turns the reference's flags + the composition of
a batch into the `mopoe_model` / `mopoe_step` descriptors of the C ABI.

Everything the reference decides in Python per batch is decided here, once per
(present modalities, N, mode) and cached:
  * the ordered powerset of modalities          (utils/BaseExperiment.py:58-79)
  * which subsets are available / fused how      (utils/BaseMMVae.py:109-134,190-216)
  * the mixture components, their weights and slice sizes
                                                 (utils/BaseMMVae.py:225-227, utils/utils.py:58-85)
  * the loss coefficients of every KL / NLL term (run_epochs.py:98-128, utils/utils.py:88-112)
Paths are relative to the reference's experiments/ directory.
"""
from collections import OrderedDict
from itertools import chain, combinations

import os
import torch

from . import _lib as L

METHODS = ("joint_elbo", "poe", "moe")


class ModelSpec:
    """Static description of a model: the flags the hot path reads
    (workflow.py:98-145).  Accepts the reference's flags object via
    ModelSpec.from_flags."""

    def __init__(self, names, input_dim, style_dim, class_dim=20,
                 method="joint_elbo", factorized=True, beta=1.0,
                 beta_style=1.0, beta_content=1.0, initial_out_logvar=-3.0,
                 learn_output_scale=True, lr=0.002, betas=(0.9, 0.999),
                 adam_eps=1e-8, rec_weights=None, poe_unimodal_elbos=True,
                 likelihood="normal", enc_layers=1, dec_layers=0, dropout=0.0,
                 sample_scale=False, gemm_operands=None):
        if method not in METHODS:
            raise NotImplementedError(
                "method %r: only joint_elbo / poe / moe are on the MI355X hot "
                "path (SURVEY.md section 8a)" % (method,))
        self.names = list(names)
        M = len(self.names)
        if not 1 <= M <= L.MAX_MODS:
            raise ValueError("1..%d modalities supported" % L.MAX_MODS)
        self.input_dim = [int(d) for d in input_dim]
        # multimodal_cohort/experiment.py:133-136
        if isinstance(style_dim, int):
            style_dim = [style_dim] * M
        elif len(style_dim) != M:
            style_dim = [style_dim[0]] * M
        # workflow.py:148-149
        self.style_dim = [int(s) for s in style_dim] if factorized else [0] * M
        self.class_dim = int(class_dim)
        self.method = method
        self.factorized = bool(factorized)
        self.beta = float(beta)
        self.beta_style = float(beta_style)
        self.beta_content = float(beta_content)
        self.initial_out_logvar = float(initial_out_logvar)
        self.learn_output_scale = bool(learn_output_scale)
        self.lr = float(lr)
        self.betas = (float(betas[0]), float(betas[1]))
        self.adam_eps = float(adam_eps)
        # run_epochs.py:115: method poe adds the unimodal ELBOs (one extra forward per
        # modality) only when this flag is set
        self.poe_unimodal_elbos = bool(poe_unimodal_elbos)
        # modalities/modality.py:18-30: what the decoder's (loc, scale) pair parameterises.
        # Normal and Laplace take that pair; Bernoulli / OneHotCategorical do not
        if likelihood not in L.LIKELIHOODS:
            raise NotImplementedError(
                "likelihood %r: the decoder's (loc, scale) output feeds Normal or Laplace"
                % (likelihood,))
        self.likelihood = likelihood
        # what the encoder layer's GEMM of a LARGE batch multiplies (mopoe_step.gemm_operands):
        # "f32" -- the reference's arithmetic, the default -- or "bf16" (opt-in, BASELINE
        # configs[1]'s "bf16 compute / fp32 accumulate": operands rounded to bfloat16, float32
        # accumulation; tests/test_oracle_precision.py measures what it costs)
        if gemm_operands is None:
            gemm_operands = os.environ.get("MOPOE_GEMM_OPERANDS", "f32")
        if gemm_operands not in ("f32", "bf16"):
            raise ValueError("gemm_operands: 'f32' or 'bf16'")
        self.gemm_operands = gemm_operands
        self.rec_weights = dict(rec_weights) if rec_weights else \
            {n: 1.0 for n in self.names}

        # BaseExperiment.set_subsets: key -> member names (sorted)
        self.subsets = OrderedDict()
        for mod_names in chain.from_iterable(
                combinations(self.names, n) for n in range(1, M + 1)):
            self.subsets["_".join(sorted(mod_names))] = sorted(mod_names)
        self.subset_keys = list(self.subsets.keys())

        # workflow.py:41-49: num_hidden_layer_encoder / num_hidden_layer_decoder /
        # dropout_rate / out_scale_per_subject (-> flags.learn_output_sample_scale, :112) --
        # networks.py:16-20,51-59.  The defaults are the fused two-launch step; any other
        # topology runs the general chain of launches (csrc/mopoe_general.inc)
        self.enc_layers = int(enc_layers)
        self.dec_layers = int(dec_layers)
        self.dropout = float(dropout)
        self.sample_scale = bool(sample_scale)
        if not (0 <= self.enc_layers <= L.MAX_LAYERS and 0 <= self.dec_layers <= L.MAX_LAYERS):
            raise ValueError("0..%d hidden layers per encoder / decoder" % L.MAX_LAYERS)
        if not 0.0 <= self.dropout < 1.0:
            raise ValueError("dropout_rate must be in [0, 1)")
        self.general = (self.enc_layers, self.dec_layers, self.dropout,
                        self.sample_scale) != (1, 0, 0.0, False) or \
            os.environ.get("MOPOE_FORCE_GENERAL") == "1"   # (experiments: the chain on the default topology)

        c = L.Model()
        c.num_mods = M
        c.class_dim = self.class_dim
        for m in range(M):
            c.input_dim[m] = self.input_dim[m]
            c.style_dim[m] = self.style_dim[m]
        c.learn_output_scale = int(self.learn_output_scale)
        t = L.Topology()
        t.enc_layers, t.dec_layers = self.enc_layers, self.dec_layers
        t.dropout, t.sample_scale = self.dropout, int(self.sample_scale)
        # (the default topology's offsets are mopoe_model_layout's)
        L.check(L.lib.mopoe_topology_layout(c, t), "mopoe_topology_layout")
        self.c_model = c
        self.c_topo = t
        self.num_floats = c.num_floats
        self._plans = {}

    @classmethod
    def from_flags(cls, flags, names):
        """Build from the SimpleNamespace workflow.train_exp assembles."""
        if getattr(flags, "modality_jsd", False):
            raise NotImplementedError("method jsd is outside the hot path")
        method = ("poe" if flags.modality_poe else
                  "moe" if flags.modality_moe else "joint_elbo")
        return cls(names, flags.input_dim, flags.style_dim,
                   class_dim=flags.class_dim, method=method,
                   factorized=flags.factorized_representation,
                   beta=flags.beta, beta_style=flags.beta_style,
                   beta_content=flags.beta_content,
                   initial_out_logvar=flags.initial_out_logvar,
                   learn_output_scale=flags.learn_output_scale,
                   lr=getattr(flags, "initial_learning_rate", 0.002),
                   betas=(getattr(flags, "beta_1", 0.9),
                          getattr(flags, "beta_2", 0.999)),
                   poe_unimodal_elbos=getattr(flags, "poe_unimodal_elbos", True),
                   likelihood=getattr(flags, "likelihood", "normal"),
                   enc_layers=getattr(flags, "num_hidden_layer_encoder", 1),
                   dec_layers=getattr(flags, "num_hidden_layer_decoder", 0),
                   dropout=getattr(flags, "dropout_rate", 0.0),
                   sample_scale=getattr(flags, "learn_output_sample_scale", False))

    @property
    def num_mods(self):
        return len(self.names)

    def has_style(self, m):
        return self.factorized and self.style_dim[m] > 0

    def heads_dim(self, m):
        return 2 * self.style_dim[m] + 2 * self.class_dim

    def z_dim(self, m):
        return self.style_dim[m] + self.class_dim

    def ldz(self, m):
        return (self.z_dim(m) + 3) // 4 * 4

    def enc_width(self, m):
        """columns of what the encoder heads read"""
        return L.HIDDEN if self.enc_layers > 0 else self.input_dim[m]

    def dec_width(self, m):
        """columns of what out_mu (and the logvar head) read"""
        return L.HIDDEN if self.dec_layers > 0 else self.z_dim(m)

    # ------------------------------------------------------------------
    def param_views(self, flat):
        """name (reference state_dict key) -> view into the flat buffer."""
        c, t = self.c_model, self.c_topo
        D, H = self.class_dim, L.HIDDEN
        out = OrderedDict()
        for m, name in enumerate(self.names):
            d, s = self.input_dim[m], self.style_dim[m]
            e = "encoders.%s." % name
            for l in range(self.enc_layers):   # nn.Sequential of (Linear, ReLU, Dropout)
                k, w0, b0 = (d if l == 0 else H), t.off_we[m][l], t.off_be[m][l]
                out[e + "shared_encoder.%d.weight" % (3 * l)] = flat[w0:w0 + H * k].view(H, k)
                out[e + "shared_encoder.%d.bias" % (3 * l)] = flat[b0:b0 + H]
            wh, bh = c.off_wh[m], c.off_bh[m]
            ew = self.enc_width(m)

            def rows(r0, n, wh=wh, ew=ew):
                return flat[wh + r0 * ew:wh + (r0 + n) * ew].view(n, ew)

            out[e + "class_mu.weight"] = rows(2 * s, D)
            out[e + "class_mu.bias"] = flat[bh + 2 * s:bh + 2 * s + D]
            out[e + "class_logvar.weight"] = rows(2 * s + D, D)
            out[e + "class_logvar.bias"] = flat[bh + 2 * s + D:bh + 2 * s + 2 * D]
            if self.has_style(m):
                out[e + "style_mu.weight"] = rows(0, s)
                out[e + "style_mu.bias"] = flat[bh:bh + s]
                out[e + "style_logvar.weight"] = rows(s, s)
                out[e + "style_logvar.bias"] = flat[bh + s:bh + 2 * s]
        for m, name in enumerate(self.names):
            d, zd, dw = self.input_dim[m], self.z_dim(m), self.dec_width(m)
            k = "decoders.%s." % name
            if not self.sample_scale:
                out[k + "logvar"] = flat[c.off_lvo[m]:c.off_lvo[m] + d].view(1, d)
            for l in range(self.dec_layers):
                kk, w0, b0 = (zd if l == 0 else H), t.off_wg[m][l], t.off_bg[m][l]
                out[k + "shared_decoder.%d.weight" % (3 * l)] = flat[w0:w0 + H * kk].view(H, kk)
                out[k + "shared_decoder.%d.bias" % (3 * l)] = flat[b0:b0 + H]
            out[k + "out_mu.weight"] = \
                flat[c.off_wd[m]:c.off_wd[m] + d * dw].view(d, dw)
            out[k + "out_mu.bias"] = flat[c.off_bd[m]:c.off_bd[m] + d]
            if self.sample_scale:     # networks.py:58-59: a Linear head instead of the parameter
                out[k + "logvar.weight"] = flat[t.off_wlv[m]:t.off_wlv[m] + d * dw].view(d, dw)
                out[k + "logvar.bias"] = flat[t.off_blv[m]:t.off_blv[m] + d]
        return out

    # ------------------------------------------------------------------
    def plan(self, present, n, sample=True, use_expert=None, backward=False,
             loss=False, group_rows=0, loss_scale=1.0):
        """`loss`: also run the decoder passes that only the loss needs (the
        unimodal forwards of method poe, run_epochs.py:104-128).  `group_rows`:
        the n rows are n/group_rows independent batches (mopoe_step.group_rows).
        `loss_scale`: weight of this batch's loss terms -- n_r * world / n_global for a
        data-parallel rank whose share of the global batch is ragged (parallel.py)."""
        key = (tuple(present), int(n), bool(sample), use_expert, bool(backward),
               bool(loss or backward), int(group_rows), float(loss_scale))
        p = self._plans.get(key)
        if p is None:
            p = StepPlan(self, *key)
            self._plans[key] = p
        return p


def _uniform_slice(n, k):
    """int(floor(N * w_0)) in the reference's float32 tensor arithmetic, with
    w = reweight_weights((1/float(K)) * ones(K))  (BaseMMVae.py:207,225;
    utils/utils.py:58-60,79)."""
    w = (1 / float(k)) * torch.ones(k)
    w = w / w.sum()
    return int(torch.floor(n * w[0])), [float(v) for v in w]


class StepPlan:
    """One `mopoe_step` descriptor + the bookkeeping to read results back."""

    def __init__(self, spec, present, n, sample, use_expert, backward, loss,
                 group_rows=0, loss_scale=1.0):
        self.spec = spec
        self.n = n
        if group_rows and (n % group_rows or backward):
            raise ValueError("group_rows must divide n and is forward-only")
        self.group_rows = group_rows
        n_sel = group_rows or n     # the batch the row-position rules see
        self.sample = sample
        self.backward = backward
        names = spec.names
        for name in present:
            if name not in names:
                raise KeyError("unknown modality %r" % (name,))
        if not present:
            raise ValueError("empty batch")
        self.present = list(present)          # batch key order
        self.present_idx = [m for m, nm in enumerate(names) if nm in present]
        mask = sum(1 << m for m in self.present_idx)
        M = spec.num_mods
        st = L.Step()
        st.n = n
        st.present_mask = mask
        st.sample = int(sample)
        st.backward = int(backward)
        st.rows_per_group = L.rows_per_group()
        st.group_rows = int(group_rows)
        st.num_subsets = len(spec.subset_keys)
        st.likelihood = L.LIKELIHOODS[spec.likelihood]
        st.gemm_operands = 1 if spec.gemm_operands == "bf16" else 0

        # subsets (BaseMMVae.inference :190-216)
        self.avail_keys = []
        avail_idx = []
        for s, (key, members) in enumerate(spec.subsets.items()):
            smask = sum(1 << names.index(nm) for nm in members)
            st.sub_mask[s] = smask
            ok = all(nm in present for nm in members)
            st.sub_avail[s] = int(ok)
            for j, nm in enumerate(members):
                st.sub_members[s][j] = names.index(nm)
            E = len(members)
            if spec.method == "moe":
                st.sub_kind[s] = L.SUB_SLICES
                st.sub_f[s] = _uniform_slice(n_sel, E)[0]
            elif spec.method == "poe" or E == M:   # BaseMMVae.poe_fusion :110-111
                st.sub_kind[s] = L.SUB_POE_PRIOR
            else:
                st.sub_kind[s] = L.SUB_POE
            if ok:
                self.avail_keys.append(key)
                avail_idx.append(s)
        self.avail_idx = avail_idx

        # mixture components (fusion_condition_* :125-134)
        if spec.method == "moe":
            comp = [s for s in avail_idx if len(spec.subsets[spec.subset_keys[s]]) == 1]
        elif spec.method == "poe":
            comp = [s for s in avail_idx
                    if len(spec.subsets[spec.subset_keys[s]]) == len(present)]
        else:
            comp = list(avail_idx)
        K = len(comp)
        f, w = _uniform_slice(n_sel, K)
        st.num_comp = K
        st.comp_f = f
        for k, s in enumerate(comp):
            st.comp_sub[k] = s
            st.comp_w[k] = w[k]
        self.comp_idx = comp
        self.comp_w = w

        if use_expert is not None:
            if use_expert not in self.avail_keys:
                raise KeyError(use_expert)     # as distr_subsets[use_expert] would
            st.joint_mode = L.JOINT_EXPERT
            st.expert_subset = spec.subset_keys.index(use_expert)
        else:
            st.joint_mode = L.JOINT_MIXTURE if sample else L.JOINT_MEAN

        # decoder jobs + the order in which the reference draws eps
        jobs = []          # (modality index, slot, src subset or -1, pass)
        for m in self.present_idx:
            jobs.append((m, 0, -1, 0))
        unimodal = spec.method == "poe" and spec.poe_unimodal_elbos
        if unimodal and loss:
            for p, nm in enumerate(self.present):     # run_epochs.py:107 order
                m = names.index(nm)
                jobs.append((m, 1, spec.subset_keys.index(nm), 1 + p))
        if len(jobs) > L.MAX_JOBS:
            raise ValueError("too many decoder jobs")
        st.num_jobs = len(jobs)
        for j, (m, slot, src, pas) in enumerate(jobs):
            st.job_mod[j] = m
            st.job_slot[j] = slot
            st.job_src[j] = src
            st.job_stream[j] = pas
            st.job_nll_coef[j] = spec.rec_weights[names[m]] if pas == 0 else 1.0
        self.jobs = jobs
        self.jobs_per_mod = [sum(1 for jb in jobs if jb[0] == m) for m in range(M)]
        # noise tape order (BaseMMVae.forward :143,155-159, one group per call)
        self.noise_slots = []
        if sample:
            passes = sorted(set(jb[3] for jb in jobs))
            for pas in passes:
                first = next(j for j, jb in enumerate(jobs) if jb[3] == pas)
                self.noise_slots.append(("content", first))
                for j, jb in enumerate(jobs):
                    if jb[3] == pas and spec.has_style(jb[0]):
                        self.noise_slots.append(("style", j))

        # loss coefficients
        b, bs, bc = spec.beta, spec.beta_style, spec.beta_content
        if spec.method in ("joint_elbo", "moe"):     # run_epochs.py:95-103
            for k, s in enumerate(comp):
                st.sub_kl_coef[s] = b * bc * w[k]
            for m in self.present_idx:
                if spec.has_style(m):
                    st.style_kl_coef[m] = b * bs * bs
        else:                                        # run_epochs.py:104-128
            coef = [0.0] * len(spec.subset_keys)
            coef[comp[0]] += b * bc                  # elbo_joint: K = 1, w = 1
            if unimodal:
                for nm in self.present:              # unimodal elbos use klds[m]
                    coef[spec.subset_keys.index(nm)] += b * bc
            for s, v in enumerate(coef):
                st.sub_kl_coef[s] = v
            for m in self.present_idx:               # style KL: joint elbo (+ unimodal one)
                if spec.has_style(m):
                    st.style_kl_coef[m] = (2.0 if unimodal else 1.0) * b * bs * bs
        self.loss_scale = float(loss_scale)
        if self.loss_scale != 1.0:
            for s in range(L.MAX_SUBSETS):
                st.sub_kl_coef[s] *= self.loss_scale
            for m in range(L.MAX_MODS):
                st.style_kl_coef[m] *= self.loss_scale
            for j in range(L.MAX_JOBS):
                st.job_nll_coef[j] *= self.loss_scale
        self.c_step = st

    def lds_bytes(self):
        return L.lib.mopoe_latent_lds_bytes(self.spec.c_model, self.c_step)

    def row_groups(self):
        """Row groups (= partial slabs) of the fused per-sample kernel for this step."""
        return L.lib.mopoe_row_groups(self.spec.c_model, self.c_step)
