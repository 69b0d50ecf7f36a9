"""XgmiComm: the gradient exchange of the data-parallel step over xGMI peer
windows (mopoe_comm_* of include/mopoe_hip.h).

One launch per rank and step pushes the flat gradient buffer to every peer
over its point-to-point link, sums the copies in rank order and applies Adam
(csrc/mopoe_xgmi.inc).  The IPC handles of the windows travel through the
process group the caller already has (any backend: all_gather_object)."""
import ctypes as C

import torch
import torch.distributed as dist

from . import _lib as L


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
        exp_avg / exp_avg_sq: `all_reduce(grads); adam_step(1/world)` in one launch."""
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
