"""Data-parallel replicas: one process per GPU, identical models, one exchange of the
flat float32 gradient buffer per step (torch.distributed: backend "nccl" = RCCL over
xGMI on the GPU node, "gloo" in the CPU tests).

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
    process only in expectation when the mixture has more than one component.

The exchange is RCCL's all-reduce by default.  The peer-window exchange of comm.py
(xGMI, rank-ordered sums) is opt-in: it has only ever run between processes that share
one GPU."""
import torch
import torch.distributed as dist


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

    exchange = "allreduce" (default): torch.distributed all_reduce of engine.grads
        (RCCL over xGMI on the node) + mopoe_adam_step(world).
    exchange = "xgmi": `comm` (an XgmiComm) pushes the buffer to every peer's window,
        sums in rank order and applies Adam in ONE launch after the backward;
    exchange = "xgmi_in_backward": the weight-gradient launch itself exchanges every
        gradient block (the N-rank step has the two launches of the 1-rank step).
    Every form refuses the update when the ranks' batches held different modalities."""

    FORMS = ("allreduce", "xgmi", "xgmi_in_backward")

    def __init__(self, engine, comm=None, exchange=None):
        if exchange is None:
            exchange = "allreduce" if comm is None else "xgmi"
        if exchange not in self.FORMS:
            raise ValueError("exchange must be one of %s" % (self.FORMS,))
        if exchange != "allreduce" and comm is None:
            raise ValueError("exchange %r needs an XgmiComm" % exchange)
        self.engine = engine
        self.comm = comm
        self.exchange = exchange
        broadcast_parameters(engine.params)
        broadcast_parameters(engine.exp_avg)
        broadcast_parameters(engine.exp_avg_sq)
        broadcast_parameters(engine.counters)     # step numbers, Adam counts

    def __call__(self, batch, eps=None, row_index=None, loss_scale=1.0, stats_host=None):
        eng = self.engine
        kw = dict(eps=eps, row_index=row_index, loss_scale=loss_scale, stats_host=stats_host)
        if world_size() == 1:
            return eng.train_step(batch, apply_adam=True, **kw)
        if self.exchange == "xgmi_in_backward":
            return eng.train_step(batch, apply_adam=True, comm=self.comm, **kw)
        out = eng.train_step(batch, apply_adam=False, **kw)
        if self.exchange == "xgmi":
            self.comm.allreduce_adam(eng)
        else:
            eng.adam_step(world=allreduce_sum_(eng.grads))
        return out
