"""CPU, world_size 2 over gloo: the data-parallel wrapper (parallel.py).  The
engine is replaced by a stand-in that produces the ORACLE's gradients for the
rank's shard (the oracle is the checker; the HIP engine needs a GPU), so the
test pins the collective logic: broadcast at start, one flat all-reduce,
1/world scaling, identical Adam updates on every rank, and equivalence with a
single process on the concatenated batch."""
import os
import socket
from collections import OrderedDict
from importlib import import_module

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import mopoe_oracle as mo

parallel = import_module("2022_cambroise_interpret_multivae_amd.parallel")


class OracleEngine:
    """Duck-typed MoPoEEngine: flat params / grads + oracle math."""

    def __init__(self, cfg, seed):
        self.cfg = cfg
        named = mo.init_params(cfg, seed)
        self.shapes = OrderedDict((k, v.shape) for k, v in named.items())
        self.params = torch.cat([v.reshape(-1) for v in named.values()])
        self.grads = torch.zeros_like(self.params)
        self.exp_avg = torch.zeros_like(self.params)
        self.exp_avg_sq = torch.zeros_like(self.params)
        self.t = 0

    def named(self, flat):
        out, o = OrderedDict(), 0
        for k, shp in self.shapes.items():
            n = int(torch.Size(shp).numel())
            out[k] = flat[o:o + n].view(shp)
            o += n
        return out

    def train_step(self, batch, eps=None, apply_adam=True):
        out, grads = mo.loss_and_grads(self.named(self.params), self.cfg, batch,
                                       mo.Noise(tape=eps))
        self.grads.zero_()
        for k, g in grads.items():
            self.named(self.grads)[k].copy_(g)
        if apply_adam:
            self.adam_step()
        return out

    def adam_step(self, present_mask=None, grad_scale=1.0):
        state = {"step": self.t, "exp_avg": self.named(self.exp_avg),
                 "exp_avg_sq": self.named(self.exp_avg_sq)}
        grads = OrderedDict((k, g * grad_scale) for k, g in self.named(self.grads).items())
        mo.adam_step(self.cfg, self.named(self.params), grads, state)
        self.t = state["step"]


def _case(method):
    cfg = mo.Config(["clinical", "rois"], [7, 444], [3, 20], method=method)
    x = mo.make_inputs(cfg.names, cfg.input_dim, 32, seed=9)
    # K = 1 mixtures only (no position-dependent slice assignment)
    present = cfg.names if method == "poe" else ["rois"]
    x = OrderedDict((k, v) for k, v in x.items() if k in present)
    noise = mo.Noise(generator=mo.noise_rng(5))
    mo.loss_and_grads(mo.init_params(cfg, 0), cfg, x, noise)
    return cfg, x, noise.tape


def _worker(rank, world, port, method, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg, x, tape = _case(method)
    n = 32 // world
    shard = OrderedDict((k, v[rank * n:(rank + 1) * n]) for k, v in x.items())
    eps = [e[rank * n:(rank + 1) * n] for e in tape]
    eng = OracleEngine(cfg, seed=rank)            # ranks start different ...
    step = parallel.DataParallelStep(eng)         # ... and are made identical
    for _ in range(2):
        step(shard, eps=eps)
    ret[rank] = eng.params.clone()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("method", ["joint_elbo", "poe"])
def test_two_ranks_equal_one_rank_on_the_full_batch(method):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(2, port, method, ret), nprocs=2, join=True)
    assert torch.equal(ret[0], ret[1])            # replicas stay identical
    cfg, x, tape = _case(method)
    single = OracleEngine(cfg, seed=0)
    for _ in range(2):
        single.train_step(x, eps=tape, apply_adam=True)
    # mean of per-shard gradients (each /N_local) == full-batch gradient (/N)
    diff = (ret[0] - single.params).abs().max().item()
    assert diff < 5e-6, diff


def test_single_process_path_needs_no_process_group():
    cfg, x, tape = _case("joint_elbo")
    eng = OracleEngine(cfg, seed=0)
    parallel.DataParallelStep(eng)(x, eps=tape)
    assert eng.t == 1 and parallel.allreduce_mean_(eng.grads) == 1.0
