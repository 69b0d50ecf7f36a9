"""MoPoEEngine: owns the device buffers of one model and drives the C ABI.

This is the host side of the hot path: flat parameter / gradient / Adam-state
buffers (one contiguous float32 allocation each, so the data-parallel
all-reduce is a single collective), per-batch-size workspaces, and the three
calls `forward`, `train_step`, `adam_step`.  All compute happens in
libmopoe_hip.so; there is no PyTorch fallback.
"""
import ctypes as C
from collections import OrderedDict

import torch

from . import _lib as L
from .plan import ModelSpec


class IndexBatch:
    """A batch as the kernels take it: `x` = {modality: the resident block (rows, d_m)} of
    the modalities every sample has, `n` rows, `row_ptr` = {modality: DEVICE ADDRESS of the n
    int32 block rows} (mopoe_buffers.row_index), `keep` = whatever owns that memory.  What
    ResidentCohort.epoch_schedule hands the loop: engine.train_step / forward take it as the
    `batch` argument without building a tensor per batch."""
    __slots__ = ("x", "n", "row_ptr", "keep")

    def __init__(self, x, n, row_ptr, keep):
        self.x, self.n, self.row_ptr, self.keep = x, n, row_ptr, keep

    def keys(self):
        return self.x.keys()

    def __contains__(self, name):
        return name in self.x

    def __len__(self):
        return len(self.x)

    def __iter__(self):
        return iter(self.x)

    def __getitem__(self, name):
        return self.x[name]

    def items(self):
        return self.x.items()

    def row_index(self):
        """{modality: (n,) int32 device tensor}: the gather vectors as tensors (tests, the
        slower paths)."""
        dev = self.keep[0]
        out = {}
        for m, p in self.row_ptr.items():
            s = (p - dev[m].data_ptr()) // 4
            out[m] = dev[m][s:s + self.n]
        return out


class Workspace:
    """Caller-owned buffers of one (batch size, jobs-per-modality) shape."""

    def __init__(self, spec, n, slots, device, backward):
        f = dict(dtype=torch.float32, device=device)
        M, D, S = spec.num_mods, spec.class_dim, len(spec.subset_keys)
        self.n = n
        self.slots = slots
        self.hidden = [torch.empty(n, L.HIDDEN, **f) for _ in range(M)]
        self.heads = [torch.empty(n, spec.heads_dim(m), **f) for m in range(M)]
        self.subsets_mu = torch.empty(S, n, D, **f)
        self.subsets_logvar = torch.empty(S, n, D, **f)
        self.joint_mu = torch.empty(n, D, **f)
        self.joint_logvar = torch.empty(n, D, **f)
        self.z = [torch.zeros(slots * n, spec.ldz(m), **f) for m in range(M)]
        self.loc = [torch.empty(slots * n, spec.input_dim[m], **f)
                    for m in range(M)]
        self._stats_all = torch.zeros(L.STATS_ALLOC, **f)   # [128:] diagnostic stamps
        self.stats = self._stats_all[:L.NUM_STATS]
        self._f = f
        self._stride = L.lib.mopoe_partials_stride(spec.c_model)
        self.partials = torch.zeros((n + L.ROWS - 1) // L.ROWS, self._stride, **f)
        self.backward = backward
        self.wgrad_scratch = None   # partial weight-gradient tiles of a large batch (train_step)
        self._cbuf = self._cbuf_partials = None   # the engine's cached mopoe_buffers
        if backward:
            self.g_xhat = [torch.empty(slots * n, spec.input_dim[m], **f)
                           for m in range(M)]
            self.g_heads = [torch.empty(n, spec.heads_dim(m), **f)
                            for m in range(M)]
            self.g_pre = [torch.empty(n, L.HIDDEN, **f) for m in range(M)]

    def ensure_partials(self, groups):
        """`partials` holds one slab per row group of the step (plan.row_groups()),
        zeroed when (re)allocated: its spare words are the fused launch's hand-off flags."""
        if self.partials.shape[0] < groups:
            self.partials = torch.zeros(groups, self._stride, **self._f)


class GeneralWorkspace(Workspace):
    """The buffers of a general topology (plan.ModelSpec.general: hidden encoder / decoder
    layers, dropout, the per-subject scale head -- mopoe_gbuffers of the C ABI).  Encoder-
    side tensors have `eb` row blocks (2 when method poe's unimodal passes re-run the
    encoder under dropout), decoder-side ones `slots`."""

    def __init__(self, spec, n, slots, device, backward, eb):
        super().__init__(spec, n, slots, device, backward)
        f = self._f
        M, H = spec.num_mods, L.HIDDEN
        self.eb = eb
        self.hidden = [None] * M            # (the default topology's single hidden layer)
        self.heads = [torch.empty(eb * n, spec.heads_dim(m), **f) for m in range(M)]
        self.partials = torch.zeros((n + L.ROWS - 1) // L.ROWS, self._stride, **f)
        self.enc_act = [[torch.empty(eb * n, H, **f) for _ in range(spec.enc_layers)]
                        for _ in range(M)]
        self.dec_act = [[torch.empty(slots * n, H, **f) for _ in range(spec.dec_layers)]
                        for _ in range(M)]
        self.lv = [torch.empty(slots * n, spec.input_dim[m], **f) if spec.sample_scale else None
                   for m in range(M)]
        self.enc_pre0 = [None] * M
        self.g_enc = self.g_dec = [[] for _ in range(M)]
        self.g_lv = self.g_z = [None] * M
        if backward:
            self.g_pre = [None] * M
            self.g_heads = [torch.empty(eb * n, spec.heads_dim(m), **f) for m in range(M)]
            self.enc_pre0 = [torch.empty(n, H, **f) if spec.dropout > 0 and spec.enc_layers
                             else None for _ in range(M)]
            self.g_enc = [[torch.empty(eb * n, H, **f) for _ in range(spec.enc_layers)]
                          for _ in range(M)]
            self.g_dec = [[torch.empty(slots * n, H, **f) for _ in range(spec.dec_layers)]
                          for _ in range(M)]
            self.g_lv = [torch.empty(slots * n, spec.input_dim[m], **f) if spec.sample_scale
                         else None for m in range(M)]
            self.g_z = [torch.empty(slots * n, spec.ldz(m), **f) for m in range(M)]
        self._gbuf = None

    def ensure_partials(self, groups):
        pass        # (the general kernels always cut the batch into groups of 16 rows)

    def gbuffers(self):
        g = self._gbuf
        if g is None:
            g = L.GBuffers()
            for m in range(len(self.heads)):
                for l, t in enumerate(self.enc_act[m]):
                    g.enc_act[m][l] = L.ptr(t)
                for l, t in enumerate(self.dec_act[m]):
                    g.dec_act[m][l] = L.ptr(t)
                for l, t in enumerate(self.g_enc[m]):
                    g.g_enc[m][l] = L.ptr(t)
                for l, t in enumerate(self.g_dec[m]):
                    g.g_dec[m][l] = L.ptr(t)
                g.enc_pre0[m] = L.ptr(self.enc_pre0[m])
                g.lv[m] = L.ptr(self.lv[m])
                g.g_lv[m] = L.ptr(self.g_lv[m])
                g.g_z[m] = L.ptr(self.g_z[m])
            self._gbuf = g
        return g


class MoPoEEngine:
    def __init__(self, spec, device="cuda", seed=None):
        if not isinstance(spec, ModelSpec):
            raise TypeError("spec must be a ModelSpec")
        self.spec = spec
        self.device = torch.device(device)
        P = spec.num_floats
        self._on_gpu = self.device.type == "cuda"
        f = dict(dtype=torch.float32, device=self.device)
        self.params = torch.zeros(P, **f)
        self.grads = torch.zeros(P, **f)
        self.exp_avg = torch.zeros(P, **f)
        self.exp_avg_sq = torch.zeros(P, **f)
        # fragment-major copies of the head / decoder weights for the four-row form of the
        # fused launch: kept by the library's own updates, rebuilt here after other writers
        self.wfrag = None if spec.general else \
            torch.zeros(L.lib.mopoe_wfrag_floats(spec.c_model), **f)
        self._wfrag_version = -1
        self.counters = torch.zeros(L.COUNTERS_ALLOC, dtype=torch.int32, device=self.device)
        # pinned host mirror {steps done, invalid} the last kernel of every training step
        # writes: an invalid step is noticed without synchronising (check_valid)
        self.status_host = torch.zeros(4, dtype=torch.int32).pin_memory() \
            if self._on_gpu else torch.zeros(4, dtype=torch.int32)
        self._status = self.status_host.numpy()
        self.device = self.params.device   # with its index: cheap `is it already there` tests
        self.views = spec.param_views(self.params)
        self.grad_views = spec.param_views(self.grads)
        self.seed = int(torch.initial_seed() if seed is None else seed) & (2 ** 63 - 1)
        self._calls = 0
        self._train_calls = 0    # training steps enqueued (run_epochs._FusedLoss)
        self._ws = {}
        self._keep = None   # tensors the in-flight kernels read (x, eps)
        self.adam = L.Adam(spec.lr, spec.betas[0], spec.betas[1], spec.adam_eps)
        self.reset_parameters()

    # ------------------------------------------------------------------ params
    def reset_parameters(self, generator=None):
        """nn.Linear's default init law (kaiming_uniform(a=sqrt(5)) = U(-1/sqrt
        (fan_in), 1/sqrt(fan_in)) for weight and bias) and decoder logvar =
        initial_out_logvar (networks.py:58-62)."""
        with torch.no_grad():
            for name, v in self.views.items():
                if name.endswith(".logvar"):
                    v.fill_(self.spec.initial_out_logvar)
                    continue
                wname = name if name.endswith(".weight") else name[:-4] + "weight"
                bound = 1.0 / (self.views[wname].shape[1] ** 0.5)
                v.copy_((torch.rand(v.shape, generator=generator) * 2 - 1) * bound)

    def load_params(self, named):
        with torch.no_grad():
            for name, v in self.views.items():
                v.copy_(named[name].reshape(v.shape))

    def named_params(self):
        return OrderedDict((k, v.detach().clone()) for k, v in self.views.items())

    def step_count(self):
        """Training steps begun (synchronises)."""
        return int(self.counters[L.CTR_STEPS_BEGUN].item())

    def adam_steps(self):
        """Adam updates applied so far to each modality's parameters (torch keeps
        state['step'] per parameter: a modality that sat out a batch is a step behind)."""
        c = self.counters[L.CTR_ADAM_STEPS:L.CTR_ADAM_STEPS + self.spec.num_mods].tolist()
        return OrderedDict(zip(self.spec.names, c))

    # ------------------------------------------------------------ step validity
    def check_valid(self, sync=False):
        """Raises MopoeError when a training step could not be completed: a hand-off
        inside the fused launch or a wait of the gradient exchange ran out of its
        budget, or the ranks' batches held different modalities.  From that step on the
        kernels apply NO Adam update (parameters and moments stay at the last complete
        step, reference run_epochs.py:180-182 never applies half a step either) until
        `recover()`.  Without `sync` this reads the pinned host mirror the kernels
        write -- no device synchronisation, at most a step or two behind."""
        if sync:
            torch.cuda.synchronize(self.device)
            bad = int(self.counters[L.CTR_INVALID].item())
        else:
            bad = int(self._status[1])      # (a numpy view of the pinned mirror: no tensor op)
        if bad:
            raise L.MopoeError(
                "a training step could not be completed (%d event(s): hand-off / "
                "gradient-exchange time-out or ranks with different modalities); no "
                "update has been applied since -- engine.recover() re-arms the step"
                % bad)

    def invalid_since(self):
        """(first step that was not applied, steps begun) after a synchronisation, or None
        when every step so far was applied.  The steps first .. begun were withheld (the
        invalid word is sticky): after `recover()` the caller runs those batches again."""
        torch.cuda.synchronize(self.device)
        c = self.counters[:4].tolist()
        if not c[L.CTR_INVALID]:
            return None
        return c[L.CTR_FIRST_INVALID], c[L.CTR_STEPS_BEGUN]

    def recover(self):
        """Re-arm after an invalid step: clears the sticky word and the hand-off flags, and
        takes the step numbers back to the last APPLIED step -- the withheld steps
        first .. begun never happened, so the caller's replay of their batches runs under
        the same step numbers, and the device noise (Philox keyed by seed and step number)
        and dropout masks of the replay are the ones an untroubled run draws: a retried run
        ends bit for bit where its twin ends (tests/test_hip_dp_onecall.py)."""
        torch.cuda.synchronize(self.device)
        c = self.counters[:4].tolist()
        if c[L.CTR_INVALID] and c[L.CTR_FIRST_INVALID] > 0:
            withheld = c[L.CTR_STEPS_BEGUN] - c[L.CTR_FIRST_INVALID] + 1
            self.counters[L.CTR_STEPS_BEGUN] = c[L.CTR_FIRST_INVALID] - 1
            self.counters[L.CTR_STEPS_DONE] = max(0, c[L.CTR_STEPS_DONE] - withheld)
        self.counters[L.CTR_INVALID] = 0
        self.counters[L.CTR_FIRST_INVALID] = 0
        for ws in self._ws.values():
            ws.partials.zero_()
        self.status_host[1] = 0
        self.status_host[2] = 0
        torch.cuda.synchronize(self.device)

    # --------------------------------------------------------------- buffers
    def workspace(self, n, slots, backward, fresh=False, eb=1):
        def make():
            if self.spec.general:
                return GeneralWorkspace(self.spec, n, slots, self.device, backward, eb)
            return Workspace(self.spec, n, slots, self.device, backward)
        if fresh:
            return make()
        key = (n, slots, backward, eb)
        ws = self._ws.get(key)
        if ws is None:
            ws = self._ws[key] = make()
        return ws

    def _bind_masks(self, plan, ws, masks, train):
        """Point the step at injected dropout keep masks (0 / 1 floats), given in the order
        the reference's Dropout modules run (networks.py:19,54): the joint forward's encoder
        stacks in modality order, its decoder stacks, then -- method poe -- one unimodal
        forward per batch modality (run_epochs.py:107).  None: the kernels draw them."""
        spec, g = self.spec, ws.gbuffers()
        M, n = spec.num_mods, plan.n
        for m in range(M):
            for l in range(L.MAX_LAYERS):
                g.keep_enc[m][l] = None
                g.keep_dec[m][l] = None
        if masks is None or not train or spec.dropout == 0.0:
            return []
        it = iter(masks)
        enc = [[[] for _ in range(spec.enc_layers)] for _ in range(M)]
        dec = [[[] for _ in range(spec.dec_layers)] for _ in range(M)]

        def take(m, stack, layers):
            for l in range(layers):
                t = next(it).to(self.device, dtype=torch.float32).contiguous()
                if tuple(t.shape) != (n, L.HIDDEN):
                    raise ValueError("keep mask of shape %s, expected %s" % (
                        tuple(t.shape), (n, L.HIDDEN)))
                stack[m][l].append(t)
        for m in plan.present_idx:
            take(m, enc, spec.enc_layers)
        for m in plan.present_idx:
            take(m, dec, spec.dec_layers)
        for (m, slot, src, pas) in plan.jobs:
            if src >= 0:            # a unimodal forward of method poe
                take(m, enc, spec.enc_layers)
                take(m, dec, spec.dec_layers)
        if next(it, None) is not None:
            raise ValueError("more keep masks than Dropout calls in this step")
        keep = []
        for m in range(M):
            for l, parts in enumerate(enc[m]):
                if parts:
                    t = torch.cat(parts[:ws.eb]) if len(parts) > 1 else parts[0]
                    keep.append(t)
                    g.keep_enc[m][l] = L.ptr(t)
            for l, parts in enumerate(dec[m]):
                if parts:
                    t = torch.cat(parts) if len(parts) > 1 else parts[0]
                    keep.append(t)
                    g.keep_dec[m][l] = L.ptr(t)
        return keep

    def refresh_wfrag(self):
        """mopoe_wfrag_refresh: the fragment-major weight copies from `params`.  Runs by
        itself whenever torch has seen a write to the flat buffer or a view of it
        (tensor._version); call it after writes torch does not count (`.data`, raw
        pointers)."""
        if self.wfrag is None:      # (a general topology: no four-row form, no copies)
            self._wfrag_version = self.params._version
            return
        b = L.Buffers()
        b.params = L.ptr(self.params)
        b.wfrag = L.ptr(self.wfrag)
        L.check(L.lib.mopoe_wfrag_refresh(self.spec.c_model, b, L.stream_ptr()),
                "mopoe_wfrag_refresh")
        self._wfrag_version = self.params._version

    def _ensure_scratch(self, plan, ws):
        """mopoe_buffers.wgrad_scratch for this step.  The count depends on the batch's
        MODALITIES as well as on its rows (a workspace serves every batch of its shape:
        clinical-only 719,360 floats, rois-only 3,013,120, both 3,668,480 at 4,096 rows), so
        it is asked per plan (cached there) and the workspace's tensor grows to the largest
        need seen; the library refuses a step whose need exceeds `wgrad_scratch_floats`.
        `ws.wgrad_scratch = False` (tests) pins the one-launch form."""
        if self.spec.general or ws.wgrad_scratch is False:
            return
        need = getattr(plan, "_scratch_need", None)
        if need is None:
            need = plan._scratch_need = int(
                L.lib.mopoe_wgrad_scratch_floats(self.spec.c_model, plan.c_step))
        have = ws.wgrad_scratch.numel() if torch.is_tensor(ws.wgrad_scratch) else 0
        if need > have:
            ws.wgrad_scratch = torch.empty(need, **ws._f)
            ws._cbuf = None

    def _optim_buffers(self, b):
        if self._on_gpu and self.params._version != self._wfrag_version:
            self.refresh_wfrag()
        b.params = L.ptr(self.params)
        b.grads = L.ptr(self.grads)
        b.exp_avg = L.ptr(self.exp_avg)
        b.exp_avg_sq = L.ptr(self.exp_avg_sq)
        b.counters = L.ptr(self.counters)
        b.status_host = L.ptr(self.status_host)
        b.wfrag = L.ptr(self.wfrag)
        return b

    def _buffers(self, ws, x, row_index, stats_host=None, plan=None):
        """The call's `mopoe_buffers`.  Everything that belongs to the workspace or the
        optimiser is filled in once per workspace (some thirty pointers); a call only sets
        its inputs, gather indices and the pinned log slot."""
        if plan is not None:
            ws.ensure_partials(plan.row_groups())
        b = ws._cbuf
        if b is None or ws._cbuf_partials is not ws.partials:
            b = L.Buffers()
            for m in range(self.spec.num_mods):
                b.hidden[m] = L.ptr(ws.hidden[m])
                b.heads[m] = L.ptr(ws.heads[m])
                b.z[m] = L.ptr(ws.z[m])
                b.loc[m] = L.ptr(ws.loc[m])
                if ws.backward:
                    b.g_xhat[m] = L.ptr(ws.g_xhat[m])
                    b.g_heads[m] = L.ptr(ws.g_heads[m])
                    b.g_pre[m] = L.ptr(ws.g_pre[m])
            b.subsets_mu = L.ptr(ws.subsets_mu)
            b.subsets_logvar = L.ptr(ws.subsets_logvar)
            b.joint_mu = L.ptr(ws.joint_mu)
            b.joint_logvar = L.ptr(ws.joint_logvar)
            b.stats = L.ptr(ws.stats)
            b.partials = L.ptr(ws.partials)
            b.wgrad_scratch = L.ptr(ws.wgrad_scratch) if torch.is_tensor(ws.wgrad_scratch) else None
            b.wgrad_scratch_floats = ws.wgrad_scratch.numel() if torch.is_tensor(ws.wgrad_scratch) else 0
            self._optim_buffers(b)
            ws._cbuf, ws._cbuf_partials = b, ws.partials
        elif self._on_gpu and self.params._version != self._wfrag_version:
            self.refresh_wfrag()
        if stats_host is not None:
            if not stats_host.is_pinned() or stats_host.numel() < L.NUM_STATS:
                raise ValueError("stats_host must be a pinned float32 tensor of >= %d"
                                 % L.NUM_STATS)
            b.stats_host = L.ptr(stats_host)
        else:
            b.stats_host = None
        for m, name in enumerate(self.spec.names):
            t = x.get(name)
            if t is not None:
                b.x[m] = L.ptr(t)
                b.x_rows[m] = t.shape[0]
                ri = row_index.get(name) if row_index is not None else None
                b.row_index[m] = None if ri is None else \
                    C.c_void_p(ri) if isinstance(ri, int) else L.ptr(ri)
            else:
                b.x[m] = None
                b.x_rows[m] = 0
                b.row_index[m] = None
        return b

    def _prepare(self, batch, row_index):
        """Device float32 inputs (run_epochs.py:85-86: .to(device).float()).
        `row_index`: None, one int tensor (n,) shared by all modalities, or a
        dict {modality: (n,) int tensor}: batch row i of modality m is row
        row_index[m][i] of batch[m] (a gather out of resident cohort arrays)."""
        L.require_gpu()
        if not self._on_gpu:
            raise L.MopoeError("engine was created on %s" % self.device)
        if isinstance(batch, IndexBatch):     # resident blocks + addresses: nothing to prepare
            return batch.x, batch.n, batch.row_ptr
        if row_index is not None and not isinstance(row_index, dict):
            row_index = {name: row_index for name in batch}
        if row_index is not None:
            row_index = {k: v.to(self.device, dtype=torch.int32).contiguous()
                         for k, v in row_index.items() if v is not None}
        x = OrderedDict()
        n = None
        for name, v in batch.items():
            t = L.device_rows(v, self.device)   # (v itself when it already qualifies)
            m = self.spec.names.index(name)
            if t.dim() != 2 or t.shape[1] != self.spec.input_dim[m]:
                raise ValueError("batch[%r] has shape %s, expected (N, %d)" % (
                    name, tuple(t.shape), self.spec.input_dim[m]))
            x[name] = t
            rows = t.shape[0]
            if row_index is not None and name in row_index:
                rows = row_index[name].shape[0]
            if n is not None and rows != n:
                raise ValueError("modalities disagree on the batch size")
            n = rows
        return x, n, row_index

    def _bind_noise(self, plan, step, eps):
        """Point the step at injected eps tensors (reference draw order)."""
        keep = []
        if eps is None and not getattr(plan, "_eps_bound", False):
            return keep          # (nothing bound since the plan was made: nothing to clear)
        for j in range(L.MAX_JOBS):
            step.job_eps_content[j] = None
            step.job_eps_style[j] = None
        plan._eps_bound = eps is not None
        if eps is None:
            return keep
        if len(eps) != len(plan.noise_slots):
            raise ValueError("expected %d eps tensors, got %d" % (
                len(plan.noise_slots), len(eps)))
        for (kind, j), e in zip(plan.noise_slots, eps):
            t = e.to(self.device, dtype=torch.float32).contiguous()
            m = plan.jobs[j][0]
            want = (plan.n, self.spec.class_dim if kind == "content"
                    else self.spec.style_dim[m])
            if tuple(t.shape) != want:
                raise ValueError("eps shape %s, expected %s" % (tuple(t.shape), want))
            keep.append(t)
            if kind == "content":
                for jj, jb in enumerate(plan.jobs):
                    if jb[3] == plan.jobs[j][3]:
                        step.job_eps_content[jj] = t.data_ptr()
            else:
                step.job_eps_style[j] = t.data_ptr()
        return keep

    # ----------------------------------------------------------------- calls
    def forward(self, batch, sample=True, use_expert=None, eps=None,
                row_index=None, loss=False, fresh=True, group_rows=0):
        """mopoe_forward: encoder, fusion, latent, decoder, loss scalars.
        `group_rows`: the batch axis holds n/group_rows independent batches."""
        x, n, row_index = self._prepare(batch, row_index)
        plan = self.spec.plan(list(x.keys()), n, sample, use_expert, False, loss,
                              group_rows)
        slots = max(plan.jobs_per_mod)
        ws = self.workspace(n, slots, False, fresh=fresh)
        step = plan.c_step
        keep = self._bind_noise(plan, step, eps)
        self._calls += 1
        step.seed = (self.seed + 0x9E3779B97F4A7C15 * self._calls) & (2 ** 64 - 1)
        self._ensure_scratch(plan, ws)   # (thousands of row groups: their pre-summed slabs)
        buf = self._buffers(ws, x, row_index, plan=plan)
        if self.spec.general:       # (evaluation: the Dropout modules are the identity)
            self._bind_masks(plan, ws, None, False)
            L.check(L.lib.mopoe_general_forward(self.spec.c_model, self.spec.c_topo, step, buf,
                                                ws.gbuffers(), L.stream_ptr()),
                    "mopoe_general_forward")
        else:
            L.check(L.lib.mopoe_forward(self.spec.c_model, step, buf, L.stream_ptr()),
                    "mopoe_forward")
        self._keep = (x, keep, row_index)
        return plan, ws

    def train_step(self, batch, eps=None, row_index=None, apply_adam=True,
                   stats_host=None, comm=None, loss_scale=1.0, rccl=None, check=True,
                   masks=None):
        """mopoe_train_step: forward + backward (+ fused Adam).  `stats_host`:
        a pinned host tensor the kernel writes the step's scalars into (the
        per-step log without a copy on the stream; read it after a sync or a
        few steps later).  `comm` (an XgmiComm): mopoe_comm_train_step -- the
        weight-gradient launch exchanges its blocks with the other ranks and
        applies Adam with the mean (all ranks: same modalities in the batch).
        `rccl` (an RcclComm): mopoe_rccl_train_step -- backward, RCCL all-reduce of the
        gradient buffer and Adam with the mean, enqueued by this one call.
        `loss_scale`: weight of the batch's loss terms (parallel.py).
        `check=False`: the caller looks after invalid steps itself (parallel.StepRetry).
        `masks`: injected dropout keep masks of a general topology (_bind_masks)."""
        if check:
            self.check_valid()      # (pinned host mirror: no synchronisation)
        x, n, row_index = self._prepare(batch, row_index)
        plan = self.spec.plan(list(x.keys()), n, True, None, True, True,
                              loss_scale=loss_scale)
        slots = max(plan.jobs_per_mod)
        step = plan.c_step
        eb = L.lib.mopoe_general_enc_blocks(self.spec.c_topo, step, 1) if self.spec.general else 1
        ws = self.workspace(n, slots, True, eb=eb)
        keep = self._bind_noise(plan, step, eps)
        step.seed = self.seed
        self._ensure_scratch(plan, ws)   # (a large batch: the split weight-gradient launches)
        buf = self._buffers(ws, x, row_index, stats_host, plan=plan)
        adam = C.byref(self.adam) if apply_adam else None
        if self.spec.general:
            if comm is not None:
                raise NotImplementedError("the peer-window exchange forms serve the default "
                                          "topology; a general one exchanges over RCCL")
            keep = keep + self._bind_masks(plan, ws, masks, True)
            L.check(L.lib.mopoe_general_train_step(
                self.spec.c_model, self.spec.c_topo, step, buf, ws.gbuffers(), adam,
                rccl._c if rccl is not None else None, L.stream_ptr()),
                "mopoe_general_train_step")
        elif comm is not None or rccl is not None:
            if not apply_adam or (comm is not None and rccl is not None):
                raise ValueError("the exchanging step applies Adam (through ONE communicator)")
            if rccl is not None:
                L.check(L.lib.mopoe_rccl_train_step(rccl._c, self.spec.c_model, step, buf,
                                                    adam, L.stream_ptr()),
                        "mopoe_rccl_train_step")
            else:
                L.check(L.lib.mopoe_comm_train_step(comm._c, self.spec.c_model, step, buf,
                                                    adam, L.stream_ptr()),
                        "mopoe_comm_train_step")
        else:
            L.check(L.lib.mopoe_train_step(self.spec.c_model, step, buf, adam,
                                           L.stream_ptr()), "mopoe_train_step")
        self._keep = (batch, x, keep, row_index)
        self._train_calls += 1
        self.last_present_mask = step.present_mask
        return plan, ws

    def adam_step(self, present_mask=None, world=1):
        """mopoe_adam_step on the flat buffers.  `world` > 1: engine.grads holds the
        SUM over `world` data-parallel ranks (after an all-reduce); the kernel applies
        the mean and checks that every rank's batch held the same modalities."""
        L.require_gpu()
        if present_mask is None:
            present_mask = self.last_present_mask
        b = self._optim_buffers(L.Buffers())
        if self.spec.general:
            L.check(L.lib.mopoe_general_adam_step(self.spec.c_model, self.spec.c_topo,
                                                  present_mask, b, C.byref(self.adam),
                                                  int(world), L.stream_ptr()),
                    "mopoe_general_adam_step")
            return
        L.check(L.lib.mopoe_adam_step(self.spec.c_model, present_mask, b,
                                      C.byref(self.adam), int(world), L.stream_ptr()),
                "mopoe_adam_step")

    def decode(self, content, styles):
        """Decoder.forward of every modality in `styles` (networks.py:66-77) on given
        latents: {name: (loc (N, d_m), scale (1, d_m))}; loc = [style | content] Wd^T +
        bd through mopoe_linear, scale = exp(logvar / 2)."""
        from . import ops
        out = OrderedDict()
        for m, name in enumerate(self.spec.names):
            if name not in styles:
                continue
            z = content
            if self.spec.has_style(m):
                z = torch.cat((styles[name].to(content.device), content), dim=1)
            k = "decoders.%s." % name
            for l in range(self.spec.dec_layers):    # (evaluation: Dropout is the identity)
                z = ops.linear(z, self.views[k + "shared_decoder.%d.weight" % (3 * l)],
                               self.views[k + "shared_decoder.%d.bias" % (3 * l)], relu=True)
            loc = ops.linear(z, self.views[k + "out_mu.weight"], self.views[k + "out_mu.bias"])
            if self.spec.sample_scale:
                logvar = ops.linear(z, self.views[k + "logvar.weight"], self.views[k + "logvar.bias"])
            else:
                logvar = self.views[k + "logvar"]
            out[name] = (loc, (logvar * 0.5).exp())
        return out

    # --------------------------------------------------------------- results
    def results(self, plan, ws):
        """The dict BaseMMVae.forward returns (BaseMMVae.py:137-165), built
        from views of the workspace (no copies)."""
        spec = self.spec
        n, D = plan.n, spec.class_dim
        enc_mods = OrderedDict()
        for m, name in enumerate(spec.names):
            s = spec.style_dim[m]
            if name in plan.present:
                h = ws.heads[m][:n]     # (a general topology's second row block: not the API's)
                enc_mods[name + "_style"] = [h[:, 0:s], h[:, s:2 * s]] \
                    if spec.has_style(m) else [None, None]
                enc_mods[name] = [h[:, 2 * s:2 * s + D], h[:, 2 * s + D:2 * s + 2 * D]]
            else:
                enc_mods[name + "_style"] = [None, None]
                enc_mods[name] = [None, None]
        subsets = OrderedDict()
        for key, s in zip(plan.avail_keys, plan.avail_idx):
            subsets[key] = [ws.subsets_mu[s], ws.subsets_logvar[s]]
        ci = plan.comp_idx
        if ci == list(range(ci[0], ci[0] + len(ci))):
            mus = ws.subsets_mu[ci[0]:ci[0] + len(ci)]
            logvars = ws.subsets_logvar[ci[0]:ci[0] + len(ci)]
        else:
            idx = torch.tensor(ci, device=self.device)
            mus = ws.subsets_mu.index_select(0, idx)
            logvars = ws.subsets_logvar.index_select(0, idx)
        latents = {
            "modalities": enc_mods, "mus": mus, "logvars": logvars,
            # BaseMMVae.py:225: the raw (1/K) weights (the reweighted ones, which
            # differ in the last bit for some K, only live inside moe_fusion and
            # calc_group_divergence_moe)
            "weights": ((1 / float(len(plan.comp_w))) *
                        torch.ones(len(plan.comp_w))).to(self.device),
            "joint": [ws.joint_mu, ws.joint_logvar], "subsets": subsets}
        res = {"latents": latents, "group_distr": latents["joint"]}
        res["joint_divergence"] = ws.stats[L.STAT_JOINT_DIV]
        if ci == list(range(ci[0], ci[0] + len(ci))):
            res["individual_divs"] = ws.stats[L.STAT_KLD_SUBSET + ci[0]:
                                              L.STAT_KLD_SUBSET + ci[0] + len(ci)]
        else:
            res["individual_divs"] = ws.stats[L.STAT_KLD_SUBSET:].index_select(
                0, torch.tensor(ci, device=self.device))
        res["dyn_prior"] = None
        rec = OrderedDict()
        for m, name in enumerate(spec.names):
            if name in plan.present:
                scale = (ws.lv[m][:n] * 0.5).exp() if spec.sample_scale else \
                    (self.views["decoders.%s.logvar" % name] * 0.5).exp()
                lik = torch.distributions.Laplace if spec.likelihood == "laplace" \
                    else torch.distributions.Normal      # (modalities/modality.py:18-30)
                rec[name] = lik(ws.loc[m][:n], scale, validate_args=False)
        res["rec"] = rec
        return res

    def log_scalars(self, plan, stats):
        """The scalar set the reference's TBLogger writes per step
        (utils/TBLogger.py:84-96: Loss, LogProb/<modality>, KLD/<subset>,
        group_divergence, mu/<latent>, logvar/<latent>) as plain floats, from ONE copy
        of the step's stats vector -- `stats` is ws.stats or the pinned host tensor the
        kernels wrote (`stats_host`); reading it costs no further synchronisation
        than the caller already paid."""
        v = stats.tolist() if hasattr(stats, "tolist") else list(stats)
        spec = self.spec
        out = OrderedDict()
        out["Loss"] = {"loss": v[L.STAT_TOTAL_LOSS]}
        out["LogProb"] = OrderedDict(
            (spec.names[m], v[L.STAT_NLL + j]) for j, (m, slot, src, pas) in
            enumerate(plan.jobs) if pas == 0)
        out["KLD"] = OrderedDict((key, v[L.STAT_KLD_SUBSET + s])
                                 for key, s in zip(plan.avail_keys, plan.avail_idx))
        out["group_divergence"] = {"group_div": v[L.STAT_JOINT_DIV]}
        mu, logvar = OrderedDict(), OrderedDict()
        for m, name in enumerate(spec.names):       # latents['modalities'] order
            if name not in plan.present:
                continue
            base = L.STAT_LATENT_MEAN + 4 * m
            if spec.has_style(m):
                mu[name + "_style"], logvar[name + "_style"] = v[base], v[base + 1]
            mu[name], logvar[name] = v[base + 2], v[base + 3]
        out["mu"], out["logvar"] = mu, logvar
        return out

    def scalars(self, plan, ws):
        """log_probs / klds / total_loss of run_epochs.basic_routine_epoch
        (run_epochs.py:89-135) as 0-dim views of the stats buffer."""
        spec = self.spec
        log_probs, klds, klds_style = OrderedDict(), OrderedDict(), OrderedDict()
        for j, (m, slot, src, pas) in enumerate(plan.jobs):
            if pas == 0:
                log_probs[spec.names[m]] = ws.stats[L.STAT_NLL + j]
        for key, s in zip(plan.avail_keys, plan.avail_idx):
            klds[key] = ws.stats[L.STAT_KLD_SUBSET + s]
        for m in plan.present_idx:
            if spec.has_style(m):
                klds_style[spec.names[m] + "_style"] = ws.stats[L.STAT_KLD_STYLE + m]
        return {"log_probs": log_probs, "klds": klds, "klds_style": klds_style,
                "total_loss": ws.stats[L.STAT_TOTAL_LOSS]}
