"""CPU, world_size 2 over gloo: the data-parallel training loop -- run_epochs.train
over a ResidentCohort, the rank-aware MissingModalitySampler and
parallel.DataParallelStep.  The engine is replaced by a stand-in that produces the
ORACLE's gradients for the rank's batch (the oracle is the checker; the HIP engine
needs a GPU) and restates the kernels' step protocol (control words of the gradient
buffer, per-modality Adam counts, the sticky invalid word), so the test pins the
loop's logic: rank 0's schedule on every rank, the same modalities on all ranks in
every step, broadcast at start, one flat all-reduce, weights of ragged shares,
identical replicas, and equivalence with ONE process on the global batches."""
import os
import socket
import types
from collections import OrderedDict
from importlib import import_module

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import mopoe_oracle as mo

_P = "2022_cambroise_interpret_multivae_amd."
parallel = import_module(_P + "parallel")
run_epochs = import_module(_P + "run_epochs")
dataset = import_module(_P + "multimodal_cohort.dataset")
L = import_module(_P + "_lib")

NAMES, DIMS, STYLE = ["clinical", "rois"], [7, 12], [3, 4]
CTRL = 8      # control words behind the parameters (the kernels use 64)


class SampleNoise(mo.Noise):
    """eps that belongs to the SAMPLE, not to its place in a batch: draw k of training
    step s for subject-row i is row i of a table seeded by (s, k), so a sample sees the
    same noise whether its batch is one rank's share or the whole global batch."""

    def __init__(self, step, rows, total_rows):
        super().__init__(generator=None)
        self.step, self.rows, self.total, self.k = step, rows, total_rows, 0

    def draw(self, shape):
        g = np.random.Generator(np.random.PCG64([977, self.step, self.k]))
        self.k += 1
        table = torch.from_numpy(g.standard_normal((self.total, shape[1])).astype(np.float32))
        return table[self.rows].to(self.dtype)


class OracleEngine:
    """Duck-typed MoPoEEngine: flat buffers + oracle math + the step protocol."""

    def __init__(self, cfg, seed, total_rows):
        self.cfg = cfg
        named = mo.init_params(cfg, seed)
        self.shapes = OrderedDict((k, v.shape) for k, v in named.items())
        flat = torch.cat([v.reshape(-1) for v in named.values()])
        self.nparam = flat.numel()
        self.params = torch.cat([flat, torch.zeros(CTRL)])
        self.grads = torch.zeros_like(self.params)
        self.exp_avg = torch.zeros_like(self.params)
        self.exp_avg_sq = torch.zeros_like(self.params)
        self.counters = torch.zeros(L.NUM_COUNTERS, dtype=torch.int32)
        self.total_rows = total_rows
        self.masks = []          # modalities of every batch this rank stepped on
        self.sizes = []

    def named(self, flat):
        out, o = OrderedDict(), 0
        for k, shp in self.shapes.items():
            n = int(torch.Size(shp).numel())
            out[k] = flat[o:o + n].view(shp)
            o += n
        return out

    def train_step(self, batch, eps=None, row_index=None, apply_adam=True, loss_scale=1.0,
                   stats_host=None, comm=None):
        self.check_valid()
        if row_index is None:       # an IndexBatch of ResidentCohort.epoch_schedule
            row_index = batch.row_index()
        x = OrderedDict((k, batch[k][row_index[k].long()]) for k in self.cfg.names if k in batch)
        first = next(iter(x))
        step = int(self.counters[L.CTR_STEPS_BEGUN]) + 1
        self.counters[L.CTR_STEPS_BEGUN] = step
        # (noise keyed by the first present modality's block row: unique per subject)
        noise = SampleNoise(step * 10 + self.cfg.names.index(first), row_index[first].long(),
                            self.total_rows)
        out, grads = mo.loss_and_grads(self.named(self.params), self.cfg, x, noise)
        self.grads.zero_()
        for k, g in grads.items():
            self.named(self.grads)[k].copy_(g * loss_scale)
        self.last_present_mask = sum(1 << m for m, k in enumerate(self.cfg.names) if k in x)
        for m in range(self.cfg.num_mods):       # control words (k_wgrad's last block)
            self.grads[self.nparam + m] = float((self.last_present_mask >> m) & 1)
        self.masks.append(self.last_present_mask)
        self.sizes.append(len(row_index[first]))
        if apply_adam:
            self.adam_step()
        return None, out

    def adam_step(self, present_mask=None, world=1):
        mask = self.last_present_mask if present_mask is None else present_mask
        ok = int(self.counters[L.CTR_INVALID]) == 0
        for m in range(self.cfg.num_mods):       # mopoe_adam_step's check of the ranks' masks
            ok &= float(self.grads[self.nparam + m]) == (world if (mask >> m) & 1 else 0)
        if not ok:
            self.counters[L.CTR_INVALID] += 1
            return
        steps = OrderedDict()
        for k in self.shapes:
            steps[k] = int(self.counters[L.CTR_ADAM_STEPS + self.cfg.names.index(k.split(".")[1])])
        state = {"step": steps, "exp_avg": self.named(self.exp_avg),
                 "exp_avg_sq": self.named(self.exp_avg_sq)}
        grads = OrderedDict((k, g / world) for k, g in self.named(self.grads).items()
                            if (mask >> self.cfg.names.index(k.split(".")[1])) & 1)
        mo.adam_step(self.cfg, self.named(self.params), grads, state)
        for m in range(self.cfg.num_mods):
            self.counters[L.CTR_ADAM_STEPS + m] += (mask >> m) & 1

    def check_valid(self, sync=False):
        if int(self.counters[L.CTR_INVALID]):
            raise L.MopoeError("a training step could not be completed")


def make_cohort():
    """41 subjects: 22 with both blocks, 12 with clinical only, 7 with rois only."""
    rng = np.random.RandomState(5)
    n = 41
    has_c = np.array([True] * 34 + [False] * 7)
    has_r = np.array([True] * 22 + [False] * 12 + [True] * 7)
    data = {"clinical": rng.randn(int(has_c.sum()), DIMS[0]),
            "rois": rng.randn(int(has_r.sum()), DIMS[1])}
    idx = {"clinical": np.array([None] * n, dtype=object), "rois": np.array([None] * n, dtype=object)}
    for name, has in (("clinical", has_c), ("rois", has_r)):
        rows = iter(range(int(has.sum())))
        for i in range(n):
            if has[i]:
                idx[name][i] = next(rows)
    return dataset.MultimodalDataset(data, idx)


def make_exp(method, batch_size, seed):
    cfg = mo.Config(NAMES, DIMS, STYLE, class_dim=5, method=method)
    ds = make_cohort()
    cohort = dataset.ResidentCohort(ds, "cpu")
    eng = OracleEngine(cfg, seed, total_rows=64)
    model = types.SimpleNamespace(engine=eng, train=lambda: None)
    exp = types.SimpleNamespace(
        flags=types.SimpleNamespace(num_models=1, batch_size=batch_size, grad_scaling=False),
        models=model, dataset_train=cohort,
        optimizers=types.SimpleNamespace(_sync=lambda: None))
    return exp, eng


def _worker(rank, world, port, method, epochs, ret, sabotage):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    exp, eng = make_exp(method, batch_size=4, seed=rank)     # ranks start different ...
    np.random.seed(100 + rank)          # ... with different numpy streams: rank 0's is used
    err = ""
    try:
        for epoch in range(epochs):
            if sabotage and rank == 1:
                # this rank's batches hold another modality set than rank 0's
                sched = exp.dataset_train.epoch_schedule
                exp.dataset_train.epoch_schedule = lambda bs, w, r: [
                    ({k: v for k, v in i.items() if k == "clinical"} if "rois" in i and
                     "clinical" in i else i, ri, wt) for i, ri, wt in sched(bs, w, r)]
            run_epochs.train(0, epoch, exp, None)            # ... and are made identical
    except L.MopoeError as e:
        err = str(e)
    ret[rank] = (eng.params.clone(), list(eng.masks), list(eng.sizes), err,
                 eng.counters.clone())
    dist.barrier()
    dist.destroy_process_group()


def _run(method, epochs, sabotage=False):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(2, port, method, epochs, ret, sabotage), nprocs=2, join=True)
    return ret


@pytest.mark.parametrize("method", ["poe", "joint_elbo"])
def test_two_ranks_through_the_training_loop(method):
    ret = _run(method, epochs=2)
    (p0, masks0, sizes0, err0, c0), (p1, masks1, sizes1, err1, c1) = ret[0], ret[1]
    assert not err0 and not err1
    assert torch.equal(p0, p1)                    # replicas stay identical
    assert masks0 == masks1                       # same modalities on all ranks, every step
    assert torch.equal(c0[:16], c1[:16])          # step numbers, per-modality Adam counts
    # both epochs visited every subject once; ragged global batches split unevenly,
    # a one-subject remainder leaves rank 1 a weight-0 share
    per_epoch = len(masks0) // 2
    assert sum(sizes0[:per_epoch]) + sum(sizes1[:per_epoch]) >= 41
    # one process on the global batches (batch_size * world), same numpy stream as rank 0
    exp, eng = make_exp(method, batch_size=8, seed=0)
    np.random.seed(100)
    for epoch in range(2):
        run_epochs.train(0, epoch, exp, None)
    assert eng.masks == masks0
    if method == "poe":     # K = 1: no position-dependent mixture slices
        diff = (p0 - eng.params).abs().max().item()
        assert diff < 2e-5, diff
    # per-modality Adam counts: clinical and rois sat out different batches
    assert int(c0[L.CTR_ADAM_STEPS]) == sum((m >> 0) & 1 for m in masks0)
    assert int(c0[L.CTR_ADAM_STEPS + 1]) == sum((m >> 1) & 1 for m in masks0)
    assert int(c0[L.CTR_ADAM_STEPS]) != int(c0[L.CTR_ADAM_STEPS + 1])


def test_ranks_with_different_modalities_are_refused():
    ret = _run("poe", epochs=1, sabotage=True)
    for r in (0, 1):
        params, masks, sizes, err, counters = ret[r]
        assert "could not be completed" in err, err
        assert int(counters[L.CTR_INVALID]) > 0
    # nothing was applied from the first mismatching step on: both replicas still agree
    assert torch.equal(ret[0][0], ret[1][0])


def test_single_process_path_needs_no_process_group():
    exp, eng = make_exp("joint_elbo", batch_size=8, seed=0)
    np.random.seed(3)
    run_epochs.train(0, 0, exp, None)
    assert int(eng.counters[L.CTR_STEPS_BEGUN]) == len(eng.masks) > 0
    assert parallel.allreduce_sum_(eng.grads) == 1


def test_shares_of_the_rank_aware_sampler():
    ds = make_cohort()
    np.random.seed(1)
    glob = dataset.MissingModalitySampler(ds, 12).draw(12)
    for world in (2, 3, 8):
        parts = [dataset.MissingModalitySampler(ds, 4, world=world, rank=r).shares(glob)
                 for r in range(world)]
        for k, b in enumerate(glob):
            got = np.concatenate([parts[r][0][k] for r in range(world)
                                  if parts[r][1][k] > 0])
            assert list(got) == list(b)                      # a partition, in order
            assert abs(sum(parts[r][1][k] for r in range(world)) - world) < 1e-9
            assert all(len(parts[r][0][k]) >= 1 for r in range(world))


# ------------------------------------------------------------ StepRetry decides collectively
class _ScriptedEngine:
    """The part of the engine StepRetry talks to, with a failure scripted per rank: the
    step numbered `fail_at` could not be completed on the ranks in `fail_ranks` (first
    attempt only); steps behind it are withheld there until recover()."""

    def __init__(self, rank, fail_at, fail_ranks):
        self.device = torch.device("cpu")
        self.status_host = torch.zeros(4, dtype=torch.int32)
        self.begun, self.first, self.recovered, self.log = 0, 0, 0, []
        self._fail = fail_at if rank in fail_ranks else None

    def run(self, tag):
        self.begun += 1
        if self._fail is not None and self.begun == self._fail and not self.recovered:
            self.first = self.begun
        self.log.append((tag, self.begun, "withheld" if self.first else "applied"))

    def invalid_since(self):
        return (self.first, self.begun) if self.first else None

    def step_count(self):
        return self.begun

    def recover(self):
        if self.first:      # (engine.recover: step numbers back to the last applied step)
            self.begun = self.first - 1
        self.first = 0
        self.recovered += 1


def _retry_worker(rank, world, port, fail_ranks, rank0_state, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    eng = _ScriptedEngine(rank, fail_at=3, fail_ranks=fail_ranks)
    recover = eng.recover
    if rank0_state:         # DataParallelStep.recover of the xgmi forms: rank 0's state everywhere
        def recover():
            eng.recover()
            t = torch.tensor([eng.begun])
            dist.broadcast(t, 0)
            eng.begun = int(t.item())
    guard = parallel.StepRetry(eng, eng.run, recover=recover, depth=8, rank0_state=rank0_state)
    err = ""
    try:
        for tag in "abcde":     # depth // 2 = 4: the ranks look after step 4 and at the flush
            guard.step(tag)
        guard.flush()
    except L.MopoeError as e:
        err = str(e)
    ret[rank] = (list(eng.log), eng.recovered, guard.retries, eng.begun, err)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("fail_ranks,rank0_state", [((0, 1), False), ((1,), False), ((1,), True),
                                                    ((0,), True)])
def test_step_retry_decides_over_all_ranks(fail_ranks, rank0_state):
    """A failure that ONE rank saw (a peer-window time-out) used to send that rank alone
    into recover() -- c10d broadcasts its peers never joined.  The look is an all-gather now:
    every rank re-arms and replays the same batches."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ret = mp.Manager().dict()
    mp.spawn(_retry_worker, args=(2, port, fail_ranks, rank0_state, ret), nprocs=2, join=True)
    (log0, rec0, tries0, begun0, err0), (log1, rec1, tries1, begun1, err1) = ret[0], ret[1]
    if not rank0_state and len(fail_ranks) == 1:
        # the all-reduce forms cannot withhold a step on one rank alone: replicas that say so
        # have come apart, and BOTH ranks raise (nobody is left waiting in a collective)
        assert "disagree" in err0 and "disagree" in err1
        assert rec0 == rec1 == 0
        return
    assert not err0 and not err1
    assert rec0 == rec1 == 1 and tries0 == tries1 == 1      # both ranks re-armed, once
    assert [t for t, _, _ in log0] == [t for t, _, _ in log1]   # ... and ran the same batches
    assert begun0 == begun1 == 5                              # five batches, five applied steps
    tags = [t for t, _, _ in log0]
    if rank0_state and 0 not in fail_ranks:
        # rank 0 applied everything: its state is broadcast, nothing is run again
        assert tags == list("abcde")
    else:
        # steps 3 and 4 were withheld (somewhere): their batches run again, as steps 3 and 4
        assert tags == list("abcd") + list("cd") + ["e"]
        assert [n for _, n, _ in log0][4:6] == [3, 4]
