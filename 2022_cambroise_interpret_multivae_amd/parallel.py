"""Data-parallel replicas: one process per GPU, identical models, one exchange of the
flat float32 gradient buffer per step (RCCL over xGMI on the GPU node; torch.distributed
"gloo" in the CPU tests).

The reference has no distributed code; the contract here is
section 8e.  A step of W ranks on per-rank batches of n_r rows is ONE step of the
reference on the global batch of sum(n_r) rows:

  * every rank's batch holds the same modalities (the rank-aware
    MissingModalitySampler deals every global batch out over the ranks), so the same
    parameters receive gradients everywhere and torch's per-parameter Adam step counts
    (reference experiment.py:256-279) stay equal on all replicas -- the Adam kernel
    verifies it through control words that ride in the gradient buffer and refuses the
    update otherwise (engine.check_valid raises);
  * a rank's loss terms are normalised by its own n_r and weighted by n_r * W / sum(n_r)
    (`loss_scale`), so the mean over ranks is the global-batch mean also for the ragged
    last batch of a modality subset;
  * the mixture slices of mixture_component_selection (utils/utils.py:63-85) are
    applied per rank (row position inside the rank's batch): equal to the single
    process only in expectation when the mixture has more than one component;
  * a step that ONE rank could not complete (a hand-off time-out in its fused launch) is
    applied by NO rank: the rank's invalid flag is another control word of the summed
    buffer, every rank raises its sticky invalid word at the same step, and StepRetry
    re-arms all of them and runs the same batches again.

The exchange is RCCL's all-reduce, by default bound inside the C ABI
(comm.RcclComm: backward, all-reduce and Adam are ONE host call per step); "allreduce"
spells the same step out over torch.distributed (any backend: the CPU tests).  The
peer-window exchange of comm.XgmiComm is opt-in: it has only ever run between processes
that share one GPU."""
import collections

import torch
import torch.distributed as dist

from . import _lib as L


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_initialized() else 0


def broadcast_parameters(flat_params, src=0):
    """Replicate rank `src`'s flat parameter buffer."""
    if world_size() > 1:
        dist.broadcast(flat_params, src)


def allreduce_sum_(flat_grads):
    """In-place sum over ranks; returns the world size (the optimiser applies 1/world,
    so the averaging costs no extra pass over the buffer)."""
    w = world_size()
    if w > 1:
        dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM)
    return w


def share_schedule(batches, src=0):
    """Rank `src`'s epoch schedule (a list of index arrays) on every rank: the ranks
    then deal out the same global batches whatever their own numpy RNG state is."""
    if world_size() == 1:
        return batches
    box = [batches if rank() == src else None]
    dist.broadcast_object_list(box, src=src)
    return box[0]


class DataParallelStep:
    """engine.train_step + gradient exchange + Adam with the mean, as one call.

    exchange = "rccl" (default on the GPU under the nccl backend): mopoe_rccl_train_step
        -- backward, ncclAllReduce of engine.grads and the Adam launch enqueued by ONE
        call of the C ABI over the library's own communicator (comm.RcclComm);
    exchange = "allreduce" (default otherwise): the same step spelled out --
        engine.train_step, torch.distributed all_reduce, engine.adam_step(world);
    exchange = "xgmi": `comm` (an XgmiComm) pushes the buffer to every peer's window and
        sums in rank order in one launch after the backward, then the Adam launch;
    exchange = "xgmi_in_backward": the weight-gradient launch itself exchanges every
        gradient block, then the Adam launch.
    Every form refuses the update when the ranks' batches held different modalities."""

    FORMS = ("rccl", "allreduce", "xgmi", "xgmi_in_backward")

    def __init__(self, engine, comm=None, exchange=None):
        if exchange is None:
            if comm is not None:
                exchange = "xgmi"
            elif (world_size() > 1 and getattr(engine, "_on_gpu", False)
                  and dist.get_backend() == "nccl"):
                exchange = "rccl"
            else:
                exchange = "allreduce"
        if exchange not in self.FORMS:
            raise ValueError("exchange must be one of %s" % (self.FORMS,))
        if exchange.startswith("xgmi") and comm is None:
            raise ValueError("exchange %r needs an XgmiComm" % exchange)
        self.engine = engine
        self.comm = comm
        self.exchange = exchange
        self.rccl = None
        if exchange == "rccl":
            from .comm import RcclComm
            self.rccl = comm if isinstance(comm, RcclComm) else RcclComm()
        self.synchronise_replicas()

    def synchronise_replicas(self, src=0):
        """Rank `src`'s parameters, moments and step counts on every rank.  c10d
        collectives do not bump tensor._version: the fragment-major weight copies the
        four-row form reads are rebuilt explicitly."""
        eng = self.engine
        broadcast_parameters(eng.params, src)
        broadcast_parameters(eng.exp_avg, src)
        broadcast_parameters(eng.exp_avg_sq, src)
        broadcast_parameters(eng.counters, src)     # step numbers, Adam counts
        if getattr(eng, "_on_gpu", False):
            eng.refresh_wfrag()

    def recover(self):
        """Re-arm every rank after an invalid step.  The RCCL / all-reduce forms withhold
        the step on every rank together (control words), so re-arming is enough.  A
        time-out of a peer-window form is seen by the rank that waited only: every rank
        holds a whole step, but the ranks may be one step apart -- rank 0's state is
        broadcast again."""
        self.engine.recover()
        if self.exchange.startswith("xgmi"):
            self.synchronise_replicas()

    def __call__(self, batch, eps=None, row_index=None, loss_scale=1.0, stats_host=None,
                 check=True, masks=None):
        eng = self.engine
        kw = dict(eps=eps, row_index=row_index, loss_scale=loss_scale, stats_host=stats_host)
        if hasattr(eng, "invalid_since"):
            kw["check"] = check
        if masks is not None:       # injected dropout keep masks (a general topology)
            kw["masks"] = masks
        if world_size() == 1 and self.rccl is None:
            return eng.train_step(batch, apply_adam=True, **kw)
        if self.exchange == "rccl":
            return eng.train_step(batch, apply_adam=True, rccl=self.rccl, **kw)
        if self.exchange == "xgmi_in_backward":
            return eng.train_step(batch, apply_adam=True, comm=self.comm, **kw)
        out = eng.train_step(batch, apply_adam=False, **kw)
        if self.exchange == "xgmi":
            self.comm.allreduce_adam(eng)
        else:
            eng.adam_step(world=allreduce_sum_(eng.grads))
        return out


class StepRetry:
    """The loop's policy for a step that could not be completed (reference
    run_epochs.py:180-182 applies a step or nothing): the kernels withhold the update from
    the first invalid step on (sticky word); this notices, re-arms and runs the withheld
    batches ONCE more, and raises if that fails too.

    `run(*args, **kw)` enqueues one training step.  One process: the pinned host mirror
    the kernels write is read after every step (no synchronisation).  Data-parallel
    replicas look synchronously every `depth // 2` steps and at `flush()` (the end of an
    epoch), and the look itself is COLLECTIVE: every rank contributes (invalid, first
    invalid step, steps begun) to an all-gather and all of them take the same branch and
    replay the same batches -- also when only one rank saw the failure (a peer-window
    time-out is seen by the rank that waited only; its peers used to walk on into the next
    exchange while it re-synchronised: mismatched collectives, a hang).  `rank0_state`: the
    recovery broadcasts rank 0's state (DataParallelStep.recover of the xgmi forms), so the
    replay starts behind RANK 0's last applied step; otherwise (RCCL / all-reduce forms:
    the control words withhold a step on every rank together) behind the earliest one.
    `depth` batches are kept for the replay.  engine.recover() takes the step numbers back
    to the last applied step, so a replayed step draws the noise of its first attempt."""

    def __init__(self, engine, run, recover=None, depth=64, world=None, rank0_state=False):
        self.engine = engine
        self.run = run
        self.recover = recover if recover is not None else engine.recover
        self.depth = int(depth)
        self.world = world_size() if world is None else world
        self.rank0_state = bool(rank0_state)
        self.history = collections.deque(maxlen=self.depth)
        # (numpy view of the pinned host mirror the kernels write: a read costs no tensor op)
        self._mirror = engine.status_host.numpy()
        self.retries = 0
        self._since = 0

    def step(self, *args, **kw):
        out = self.run(*args, **kw)
        self.history.append((args, kw))
        self._since += 1
        if self.world == 1:
            if self._mirror[1] != 0:
                out = self._redo(self.engine.invalid_since()) or out
        elif self._since >= max(1, self.depth // 2):
            out = self._redo(self._agreed()) or out
        return out

    def flush(self):
        """Nothing invalid is left behind (synchronises; collective over the replicas)."""
        self._redo(self._agreed())
        self.history.clear()

    def end_epoch(self):
        """End of an epoch.  Replicas look synchronously (they must decide together); one
        process reads the pinned mirror without waiting for the GPU -- a step that fails
        after this look is retried at the next one (`history` keeps its batch)."""
        if self.world > 1:
            self.flush()
        elif self._mirror[1] != 0:
            self._redo(self.engine.invalid_since())

    def _agreed(self):
        """(first step to run again, steps begun) -- the SAME pair on every rank -- or None
        when no rank holds an invalid step.  Synchronises; collective when world > 1."""
        self._since = 0
        info = self.engine.invalid_since()
        if self.world == 1 or not dist.is_initialized():
            return info
        begun = info[1] if info is not None else self.engine.step_count()
        dev = self.engine.device if dist.get_backend() == "nccl" else "cpu"
        mine = torch.tensor([1 if info is not None else 0, info[0] if info is not None else 0, begun],
                            dtype=torch.int64, device=dev)
        everyone = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(everyone, mine)
        table = [t.tolist() for t in everyone]
        if not any(row[0] for row in table):
            return None
        begun = max(row[2] for row in table)
        if self.rank0_state:         # rank 0's state is what every rank continues from
            first = table[0][1] if table[0][0] else table[0][2] + 1
        else:
            # the RCCL / all-reduce forms withhold a step on every rank together (the control
            # words of the summed gradient buffer): ranks that disagree here are replicas that
            # have come apart -- every rank sees the same table, so all of them raise
            if not all(row[0] for row in table) or len({row[1] for row in table}) != 1:
                raise L.MopoeError(
                    "data-parallel replicas disagree on the steps that were withheld "
                    "(invalid, first invalid step, steps begun per rank: %s)" % (table,))
            first = table[0][1]
        return first, begun

    def _redo(self, info):
        if info is None:
            return None
        first, begun = info
        k = begun - first + 1
        if first <= 0 or k > len(self.history):
            raise L.MopoeError(
                "training steps %d..%d could not be completed and only the last %d batches "
                "are kept for a retry" % (first, begun, len(self.history)))
        replay = list(self.history)[-k:] if k > 0 else []
        self.recover()
        self.retries += 1
        out = None
        for args, kw in replay:
            out = self.run(*args, **kw)
        if (self._agreed() if self.world > 1 else self.engine.invalid_since()) is not None:
            raise L.MopoeError(
                "%d training step(s) could not be completed twice in a row (hand-off or "
                "gradient-exchange time-out, or ranks with different modalities); the "
                "parameters are those of the last complete step" % k)
        self._since = 0
        return out
