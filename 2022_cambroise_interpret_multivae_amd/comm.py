"""The communicators of the data-parallel step (include/mopoe_hip.h).

RcclComm (mopoe_rccl_*, the default): the library's own RCCL communicator; its
`train_step` is the whole N-rank step -- backward, all-reduce of the flat gradient
buffer, Adam with the mean -- as ONE call (csrc/mopoe_rccl.inc).

XgmiComm (mopoe_comm_*, opt-in): one launch per rank and step pushes the flat gradient
buffer to every peer over its point-to-point link and sums the copies in rank order; the
Adam launch behind it applies the mean or nothing (csrc/mopoe_xgmi.inc).

The out-of-band data (RCCL's unique id, the IPC handles of the windows) travels through
the process group the caller already has (any backend)."""
import ctypes as C

import torch
import torch.distributed as dist

from . import _lib as L


class RcclComm:
    """ncclCommInitRank over the ranks of `group` (collective: every rank constructs
    it; the current device is the rank's GPU)."""

    def __init__(self, group=None):
        L.require_gpu()
        if not dist.is_initialized():
            raise L.MopoeError("RcclComm needs an initialised process group")
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self._c = C.c_void_p()
        buf = C.create_string_buffer(L.RCCL_ID_BYTES)
        rc, err = 0, ""
        if self.rank == 0:
            rc = L.lib.mopoe_rccl_unique_id(buf)
            err = L.lib.mopoe_last_error().decode("utf-8", "replace") if rc else ""
        box = [(rc, bytes(buf.raw), err)]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group else 0,
                                   group=group)
        rc, raw, err = box[0]
        if rc:          # (an error on every rank, not a hang on some)
            raise L.MopoeError("mopoe_rccl_unique_id failed on rank 0: %s" % err)
        rc = L.lib.mopoe_rccl_create(self.rank, self.world, raw, C.byref(self._c))
        err = L.lib.mopoe_last_error().decode("utf-8", "replace") if rc else ""
        status = [None] * self.world
        dist.all_gather_object(status, (rc, err), group=group)
        bad = [(r, e[1]) for r, e in enumerate(status) if e[0] != 0]
        if bad:
            self.close()
            raise L.MopoeError("RCCL communicator could not be set up: %s" % (bad,))

    def allreduce_(self, flat):
        """In place: flat (float32, device, contiguous) <- sum over ranks."""
        L.require_gpu(flat)
        if flat.dtype != torch.float32 or not flat.is_contiguous():
            raise ValueError("expected a contiguous float32 device tensor")
        L.check(L.lib.mopoe_rccl_allreduce(self._c, L.ptr(flat), flat.numel(), L.stream_ptr()),
                "mopoe_rccl_allreduce")
        return flat

    def close(self):
        if self._c:
            torch.cuda.synchronize()
            L.lib.mopoe_rccl_destroy(self._c)
            self._c = C.c_void_p()


class XgmiComm:
    def __init__(self, num_floats, group=None, timeout_ms=2000):
        L.require_gpu()
        if not dist.is_initialized():
            raise L.MopoeError("XgmiComm needs an initialised process group")
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.num_floats = int(num_floats)
        self._c = C.c_void_p()
        handle = C.create_string_buffer(L.IPC_HANDLE_BYTES)
        rc = L.lib.mopoe_comm_create(self.rank, self.world, self.num_floats,
                                     int(timeout_ms), C.byref(self._c), handle)
        # every rank takes part in the exchange below even if its own create
        # failed, so that a failure is an error on all ranks, not a hang on some
        mine = (rc, bytes(handle.raw),
                L.lib.mopoe_last_error().decode("utf-8", "replace") if rc else "")
        everyone = [None] * self.world
        dist.all_gather_object(everyone, mine, group=group)
        bad = [(r, e[2]) for r, e in enumerate(everyone) if e[0] != 0]
        rc = 0
        err = ""
        if not bad:
            rc = L.lib.mopoe_comm_connect(self._c, b"".join(e[1] for e in everyone))
            err = L.lib.mopoe_last_error().decode("utf-8", "replace") if rc else ""
        status = [None] * self.world
        dist.all_gather_object(status, (rc, err), group=group)
        bad += [(r, e[1]) for r, e in enumerate(status) if e[0] != 0]
        if bad:
            self.close(barrier=False)
            raise L.MopoeError("xGMI windows could not be set up: %s" % (bad,))

    def allreduce_(self, flat):
        """In place: flat (num_floats, float32, device) <- sum over ranks, added
        in rank order (the same bits on every rank)."""
        L.require_gpu(flat)
        if flat.dtype != torch.float32 or flat.numel() != self.num_floats \
                or not flat.is_contiguous():
            raise ValueError("expected a contiguous float32 tensor of %d elements"
                             % self.num_floats)
        L.check(L.lib.mopoe_comm_allreduce(self._c, L.ptr(flat), L.stream_ptr()),
                "mopoe_comm_allreduce")
        return flat

    def allreduce_adam(self, engine, present_mask=None):
        """engine.grads <- rank-ordered sum; Adam with the mean on engine.params /
        exp_avg / exp_avg_sq: `all_reduce(grads); adam_step(1/world)` in one call (the
        exchange launch + the Adam launch, which applies the whole step or none of it)."""
        if present_mask is None:
            present_mask = engine.last_present_mask
        b = engine._optim_buffers(L.Buffers())
        L.check(L.lib.mopoe_comm_allreduce_adam(
            self._c, engine.spec.c_model, present_mask, b, C.byref(engine.adam),
            L.stream_ptr()), "mopoe_comm_allreduce_adam")

    def timeouts(self):
        """Waits that ran out of their budget so far (0 = every exchange was
        complete).  Synchronises the device."""
        torch.cuda.synchronize()
        n = C.c_int32(0)
        L.check(L.lib.mopoe_comm_status(self._c, C.byref(n)), "mopoe_comm_status")
        return int(n.value)

    def close(self, barrier=True):
        if self._c:
            if barrier:
                torch.cuda.synchronize()
                dist.barrier(group=self.group)
            L.lib.mopoe_comm_destroy(self._c)
            self._c = C.c_void_p()
