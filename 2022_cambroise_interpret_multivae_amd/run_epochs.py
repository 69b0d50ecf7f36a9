"""Mirror of the reference's run_epochs.py: basic_routine_epoch / train /
test / run_epochs with the reference's signatures and return values, the hot
loop running on the HIP engine.

`basic_routine_epoch` under torch.no_grad() (test) runs the fused forward and
returns the loss terms; with grad enabled it runs the fused forward+backward
(gradients land in the model's flat gradient buffer) and returns a
`total_loss` whose `.backward()` publishes them as `param.grad`, so that the
reference's `optimizer.zero_grad(); total_loss.backward(); optimizer.step()`
sequence (run_epochs.py:180-182) is unchanged.
"""
import torch
from torch.utils.data import DataLoader

from . import _lib as L
from . import checkpoint, parallel


class _FusedLoss(torch.autograd.Function):
    """total_loss of a step whose gradients already sit in engine.grads."""

    @staticmethod
    def forward(ctx, anchor, model, loss_view):
        ctx.model = model
        ctx.step = model.engine._train_calls
        return loss_view.detach().clone()

    @staticmethod
    def backward(ctx, grad_out):
        model = ctx.model
        eng = model.engine
        # The gradients already sit in engine.grads for d(total_loss) = 1: they were
        # computed by the step itself, not by this call.  A second backward() of the same
        # loss (retain_graph) or one after a later step would scale / publish a buffer that
        # no longer belongs to this loss: refused.
        if ctx.step is None or ctx.step != eng._train_calls:
            raise RuntimeError("total_loss.backward() publishes the gradients of the step "
                               "that produced it, once: call it before the next training "
                               "step and not a second time")
        ctx.step = None
        # a caller that scales the loss before .backward() gets the gradients scaled: one
        # in-place multiply over the parameters' part of the buffer (the control words
        # behind it are left alone)
        eng.grads[:eng.spec.c_model.off_ctrl].mul_(grad_out)
        mask = eng.last_present_mask
        names = eng.spec.names
        for name, p in model.named_parameters():
            if not p.requires_grad:
                continue
            m = names.index(name.split(".")[1])
            if not (mask >> m) & 1:
                continue          # absent modality: .grad stays None (as in torch)
            g = eng.grad_views[name]
            p.grad = g if p.grad is None or p.grad.data_ptr() == g.data_ptr() else p.grad + g
        return None, None, None


class _Lazy(dict):
    """A dict filled on first use.  The training loop only reads `results`,
    `log_probs` and `klds` when it logs; building the ~40 tensor views and the
    Normal objects of BaseMMVae.forward's dict costs more host time than the step
    takes on the GPU.  (As with the eager dict, the entries are views of the step's
    workspace: read them before the next training step overwrites it.)"""

    def __init__(self, build):
        super().__init__()
        self._build = build

    def _fill(self):
        if self._build is not None:
            build, self._build = self._build, None
            dict.update(self, build())

    def __getitem__(self, k):
        self._fill()
        return dict.__getitem__(self, k)

    def __iter__(self):
        self._fill()
        return dict.__iter__(self)

    def __len__(self):
        self._fill()
        return dict.__len__(self)

    def __contains__(self, k):
        self._fill()
        return dict.__contains__(self, k)

    def __repr__(self):
        self._fill()
        return dict.__repr__(self)

    def keys(self):
        self._fill()
        return dict.keys(self)

    def values(self):
        self._fill()
        return dict.values(self)

    def items(self):
        self._fill()
        return dict.items(self)

    def get(self, k, default=None):
        self._fill()
        return dict.get(self, k, default)


def basic_routine_epoch(exp, model_idx, batch):
    """reference run_epochs.py:73-135"""
    model = exp.models
    if exp.flags.num_models > 1:
        model = model[model_idx]
    batch_d = batch[0]
    dev = model.engine.device
    for m_key in batch_d.keys():                     # run_epochs.py:85-86
        t = batch_d[m_key]
        if t.device != dev or t.dtype != torch.float32:
            batch_d[m_key] = t.to(dev).float()
    eng = model.engine
    if torch.is_grad_enabled():
        plan, ws = eng.train_step(batch_d, apply_adam=False)
        total_loss = _FusedLoss.apply(model._anchor, model, ws.stats[L.STAT_TOTAL_LOSS])
    else:
        plan, ws = eng.forward(batch_d, sample=True, loss=True, fresh=True)
        total_loss = ws.stats[L.STAT_TOTAL_LOSS]
    sc = _Lazy(lambda: eng.scalars(plan, ws))
    return {"results": _Lazy(lambda: eng.results(plan, ws)),
            "log_probs": _Lazy(lambda: sc["log_probs"]),
            "total_loss": total_loss, "klds": _Lazy(lambda: sc["klds"])}


def _log_step(tb_logger, eng, plan, ws):
    if tb_logger is not None:
        sc = eng.scalars(plan, ws)
        tb_logger.write_training_logs(eng.results(plan, ws), sc["total_loss"],
                                      sc["log_probs"], sc["klds"])


def train(model_idx, epoch, exp, tb_logger):
    """reference run_epochs.py:138-184.

    * `exp.dataset_train` a dataset.ResidentCohort: the epoch runs on index batches
      over HBM-resident blocks (no per-sample host work), one fused launch pair per
      step.  Under torch.distributed (one process per GPU) the ranks are data-parallel
      replicas: the rank-aware MissingModalitySampler deals every global batch of
      batch_size * world samples out over the ranks (all ranks: the same modalities),
      and parallel.DataParallelStep exchanges the gradients and applies Adam with the
      mean (RCCL all-reduce unless `exp.dp_exchange` / `exp.dp_comm` choose the
      peer-window forms).
    * otherwise batches come from a DataLoader exactly as in the reference and the
      step keeps the reference's shape: zero_grad / backward / optimizer.step."""
    model, dataset, optimizer = exp.models, exp.dataset_train, exp.optimizers
    if exp.flags.num_models > 1:
        model, dataset, optimizer = model[model_idx], dataset[model_idx], optimizer[model_idx]
    model.train()
    if getattr(exp.flags, "grad_scaling", False):
        raise NotImplementedError("grad_scaling (the reference's branch never calls "
                                  "zero_grad, SURVEY.md appendix C)")
    eng = model.engine
    world = parallel.world_size()
    if hasattr(dataset, "epoch_schedule"):             # ResidentCohort
        step = _dp_step(exp, model_idx, eng) if world > 1 else None
        guard = _retry_guard(exp, model_idx, eng, step)
        for inputs, row_index, weight in dataset.epoch_schedule(
                exp.flags.batch_size, world, parallel.rank()):
            optimizer._sync()
            if guard is not None:
                plan, ws = guard.step(inputs, row_index, weight)
            elif step is None:
                plan, ws = eng.train_step(inputs, row_index=row_index, apply_adam=True)
            else:
                plan, ws = step(inputs, row_index=row_index, loss_scale=weight)
            _log_step(tb_logger, eng, plan, ws)
        if guard is not None:
            # One process: no synchronisation at the end of an epoch -- the kernels' pinned
            # mirror is looked at (a step that failed late is retried at the next look: the
            # parameters are those of the last complete step meanwhile, run_epochs() looks
            # synchronously after the last epoch).  Replicas look synchronously, together.
            guard.end_epoch()
        else:
            eng.check_valid(sync=True)  # no half-applied step leaves the epoch unnoticed
        return
    if world > 1:
        raise NotImplementedError(
            "data-parallel training runs over a dataset.ResidentCohort (the DataLoader "
            "path keeps the reference's single-process zero_grad / backward / step)")
    for batch in _loader(exp, dataset, train=True):
        basic_routine = basic_routine_epoch(exp, model_idx, batch)
        optimizer.zero_grad()
        basic_routine["total_loss"].backward()
        optimizer.step()
        if tb_logger is not None:
            tb_logger.write_training_logs(basic_routine["results"],
                                          basic_routine["total_loss"],
                                          basic_routine["log_probs"], basic_routine["klds"])
    eng.check_valid(sync=True)


def _retry_guard(exp, model_idx, eng, dp_step):
    """The model's parallel.StepRetry: a training step the kernels could not complete (a
    transient co-tenant of the GPU can starve a hand-off inside the fused launch) costs one
    retried step instead of the epoch.  The reference applies a step or nothing
    (run_epochs.py:180-182); so does this."""
    if not hasattr(eng, "invalid_since"):
        return None          # (a stand-in engine without the device-side step protocol)
    guards = exp.__dict__.setdefault("_retry_guards", {})
    if model_idx not in guards:
        if dp_step is None:
            def run(inputs, row_index, weight):
                return eng.train_step(inputs, row_index=row_index, apply_adam=True, check=False)
            guards[model_idx] = parallel.StepRetry(eng, run)
        else:
            def run(inputs, row_index, weight):
                return dp_step(inputs, row_index=row_index, loss_scale=weight, check=False)
            guards[model_idx] = parallel.StepRetry(eng, run, recover=dp_step.recover,
                                                   rank0_state=dp_step.exchange.startswith("xgmi"))
    return guards[model_idx]


def _dp_step(exp, model_idx, eng):
    """The model's DataParallelStep, made once (its constructor broadcasts rank 0's
    parameters, moments and step counts)."""
    steps = exp.__dict__.setdefault("_dp_steps", {})
    if model_idx not in steps:
        steps[model_idx] = parallel.DataParallelStep(
            eng, comm=getattr(exp, "dp_comm", None),
            exchange=getattr(exp, "dp_exchange", None))
    return steps[model_idx]


def test(model_idx, epoch, exp, tb_logger):
    """reference run_epochs.py:187-219 (latents are still sampled)."""
    with torch.no_grad():
        model, dataset = exp.models, exp.dataset_test
        if exp.flags.num_models > 1:
            model, dataset = model[model_idx], dataset[model_idx]
        model.eval()
        for batch in _loader(exp, dataset, train=False):
            basic_routine = basic_routine_epoch(exp, model_idx, batch)
            if tb_logger is not None:
                tb_logger.write_testing_logs(basic_routine["results"],
                                             basic_routine["total_loss"],
                                             basic_routine["log_probs"],
                                             basic_routine["klds"])


def _loader(exp, dataset, train):
    """The reference builds DataLoader(dataset, batch_sampler=
    MissingModalitySampler(...), num_workers=8) for training and a plain
    batched DataLoader for testing (run_epochs.py:155-157,201)."""
    if hasattr(dataset, "__iter__") and not hasattr(dataset, "__getitem__"):
        return dataset                      # an iterable of ready batches
    sampler_cls = getattr(exp, "batch_sampler_cls", None)
    workers = getattr(exp.flags, "num_workers", 0)
    if train and sampler_cls is not None:
        return DataLoader(dataset, batch_sampler=sampler_cls(
            dataset, batch_size=exp.flags.batch_size), num_workers=workers)
    return DataLoader(dataset, batch_size=exp.flags.batch_size, num_workers=workers)


def run_epochs(exp, tb_logger=None):
    """reference run_epochs.py:222-256: epochs of train + test; a checkpoint every fifth
    epoch and after the last one, in the reference's layout (checkpoint.py).  Data-
    parallel replicas hold identical parameters: rank 0 writes."""
    flags = exp.flags
    many = flags.num_models > 1
    for model_idx in range(flags.num_models):
        for epoch in range(flags.start_epoch, flags.end_epoch):
            train(model_idx, epoch, exp, tb_logger)
            test(model_idx, epoch, exp, tb_logger)
            due = (epoch + 1) % 5 == 0 or epoch + 1 == flags.end_epoch
            if due:                      # (nothing half-done goes into a checkpoint)
                for guard in exp.__dict__.get("_retry_guards", {}).values():
                    guard.flush()
            if due and parallel.rank() == 0:
                checkpoint.save_model(exp.models[model_idx] if many else exp.models, flags,
                                      epoch, model_idx if many else None)
