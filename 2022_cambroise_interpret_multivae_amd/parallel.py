"""Data-parallel replicas: one process per GPU, identical models, one
all-reduce of the flat float32 gradient buffer per step (torch.distributed:
backend "nccl" = RCCL over xGMI on the GPU node, "gloo" in the CPU tests).

The reference has no distributed code; the contract here
is section 8e: per-rank batches normalised by their own N, gradients averaged,
every rank applies the same Adam update."""
import torch
import torch.distributed as dist


def broadcast_parameters(flat_params, src=0):
    """Replicate rank `src`'s flat parameter buffer."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(flat_params, src)


def allreduce_mean_(flat_grads):
    """In-place sum over ranks; returns the scale (1/world) the optimiser
    applies, so the averaging costs no extra pass over the buffer."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return 1.0
    dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM)
    return 1.0 / dist.get_world_size()


class DataParallelStep:
    """engine.train_step + gradient all-reduce + engine.adam_step.

    `comm`: an XgmiComm (comm.py) -- the exchange goes over the node's xGMI peer
    windows instead of the library all-reduce, the sum is taken in rank order
    (replicas stay bit-identical) and Adam rides in the same launch: inside the
    weight-gradient launch itself (`in_backward=True`, the default: the N-rank
    step has the two launches of the one-rank step; every rank's batch must hold
    the same modalities), or as one launch after it (`in_backward=False`: any
    mix of modality masks across ranks)."""

    def __init__(self, engine, comm=None, in_backward=True):
        self.engine = engine
        self.comm = comm
        self.in_backward = in_backward
        broadcast_parameters(engine.params)
        broadcast_parameters(engine.exp_avg)
        broadcast_parameters(engine.exp_avg_sq)

    def __call__(self, batch, eps=None):
        eng = self.engine
        world = dist.get_world_size() if dist.is_initialized() else 1
        if world == 1:
            return eng.train_step(batch, eps=eps, apply_adam=True)
        if self.comm is not None and self.in_backward:
            return eng.train_step(batch, eps=eps, apply_adam=True, comm=self.comm)
        out = eng.train_step(batch, eps=eps, apply_adam=False)
        if self.comm is not None:
            self.comm.allreduce_adam(eng)
            return out
        scale = allreduce_mean_(eng.grads)
        eng.adam_step(grad_scale=scale)
        return out
