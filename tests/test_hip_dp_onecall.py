"""GPU: the data-parallel step as ONE call of the C ABI (mopoe_rccl_train_step), the
whole-step-or-nothing rule of every exchange form, and the loop's retry policy.

* the one-call RCCL step (backward, ncclAllReduce, Adam enqueued by one host call over the
  library's own communicator) leaves the same BITS as the spelled-out step
  (mopoe_train_step(adam = NULL) -> all_reduce -> mopoe_adam_step) -- on the one GPU of
  the test box the group has one rank, so the collective is RCCL's one-rank path; the
  N-rank arithmetic (1 / world, the control words) is pinned by feeding mopoe_adam_step
  a hand-made "sum over two ranks";
* a rank whose backward could not be completed says so in a control word of the gradient
  buffer, and mopoe_adam_step then applies nothing on ANY rank;
* a peer-window exchange in which ONE block's wait fails applies nothing at all (the
  advisor's round-2 finding: the other blocks used to apply their share);
* DataParallelStep rebuilds the fragment-major weight copies after its broadcasts (c10d
  collectives do not bump tensor._version);
* run_epochs.train retries a step that could not be completed: one forced hand-off
  time-out in mid-epoch costs one retried step, and the epoch ends where a twin that never
  saw the failure ends; a second failure in a row raises."""
import os
import socket
import types
from collections import OrderedDict
from importlib import import_module

import numpy as np
import pytest
import torch
import torch.distributed as dist

import mopoe_amd as mm
import mopoe_oracle as mo
from hip_util import make_engine

pytestmark = pytest.mark.gpu
L = mm._lib
_P = "2022_cambroise_interpret_multivae_amd."
parallel = import_module(_P + "parallel")
run_epochs = import_module(_P + "run_epochs")
dataset = import_module(_P + "multimodal_cohort.dataset")
CFG = dict(names=["clinical", "rois"], input_dim=[7, 444], style_dim=[3, 20])


def _batch(cfg, n, seed, present=None):
    x = mo.make_inputs(cfg.names, cfg.input_dim, n, seed=seed)
    return OrderedDict((k, v) for k, v in x.items() if present is None or k in present)


def _eps(cfg, n, seed, present=None):
    g = mo.noise_rng(seed)
    shapes = [(n, cfg.class_dim)] + [(n, s) for k, s in zip(cfg.names, cfg.style_dim)
                                     if present is None or k in present]
    return [torch.from_numpy(g.standard_normal(s).astype(np.float32)) for s in shapes]


@pytest.fixture(scope="module")
def one_rank_group():
    """A one-rank nccl (= RCCL) process group on the test box's single GPU."""
    if dist.is_initialized():
        yield
        return
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0,
                            world_size=1, device_id=torch.device("cuda", 0))
    yield
    dist.destroy_process_group()


def test_one_call_rccl_step_leaves_the_bits_of_the_spelled_out_step(one_rank_group):
    cfg = mo.Config(**CFG)
    _, a = make_engine(cfg)
    _, b = make_engine(cfg)
    one = parallel.DataParallelStep(a, exchange="rccl")
    assert one.rccl is not None and one.rccl.world == 1
    sets = [None, ["rois"], None, ["clinical"], None]       # changing modality sets
    for k, present in enumerate(sets):
        n = 256 if present is None else 64
        x, eps = _batch(cfg, n, 30 + k, present), _eps(cfg, n, 60 + k, present)
        one(x, eps=eps)
        b.train_step(x, eps=eps, apply_adam=False)
        dist.all_reduce(b.grads)
        b.adam_step(world=1)
    torch.cuda.synchronize()
    a.check_valid(sync=True)
    for name in ("params", "exp_avg", "exp_avg_sq", "grads", "wfrag"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    assert torch.equal(a.counters[:16], b.counters[:16])
    assert a.adam_steps() == b.adam_steps() == OrderedDict(clinical=4, rois=4)
    # and it is the fused one-rank step's arithmetic too (the same update, applied by
    # k_adam instead of the weight-gradient launch's epilogue)
    _, c = make_engine(cfg)
    for k, present in enumerate(sets):
        n = 256 if present is None else 64
        c.train_step(_batch(cfg, n, 30 + k, present), eps=_eps(cfg, n, 60 + k, present))
    torch.cuda.synchronize()
    assert torch.equal(a.params, c.params) and torch.equal(a.exp_avg_sq, c.exp_avg_sq)
    # the plain collective of the same communicator
    t = torch.arange(1000, dtype=torch.float32, device="cuda")
    one.rccl.allreduce_(t)
    torch.cuda.synchronize()
    assert torch.equal(t.cpu(), torch.arange(1000, dtype=torch.float32))
    one.rccl.close()


def test_a_rank_that_could_not_finish_its_backward_stops_every_rank():
    """Control word MAX_MODS of the gradient buffer = this rank's invalid flag; the Adam
    kernel of an N-rank step requires the all-reduced flag to be zero."""
    cfg = mo.Config(**CFG)
    spec, a = make_engine(cfg)
    ctrl = spec.c_model.off_ctrl
    x, eps = _batch(cfg, 64, 5), _eps(cfg, 64, 6)
    a.train_step(x, eps=eps, apply_adam=False)
    torch.cuda.synchronize()
    assert a.grads[ctrl:ctrl + L.MAX_MODS + 1].tolist() == [1.0, 1.0, 0.0, 0.0, 0.0, 0.0]
    # "sum over two ranks" whose second rank raised its flag: nothing is applied HERE
    before = a.params.clone()
    a.grads.mul_(2.0)
    a.grads[ctrl + L.MAX_MODS] = 1.0
    a.adam_step(world=2)
    torch.cuda.synchronize()
    assert torch.equal(before, a.params)
    assert a.adam_steps() == OrderedDict(clinical=0, rois=0)
    assert a.invalid_since() == (1, 1)               # step 1 was withheld
    with pytest.raises(L.MopoeError):
        a.check_valid(sync=True)
    # a rank whose own hand-off timed out raises the flag itself
    _, b = make_engine(cfg)
    os.environ["MOPOE_TEST_HANDOFF_SPINS"] = "0"
    L.reload_knobs()
    try:
        b.train_step(_batch(cfg, 256, 7), eps=_eps(cfg, 256, 8), apply_adam=False)
        torch.cuda.synchronize()
    finally:
        del os.environ["MOPOE_TEST_HANDOFF_SPINS"]
        L.reload_knobs()
    assert b.grads[ctrl + L.MAX_MODS].item() == 1.0
    assert b.invalid_since() == (1, 1)
    assert b.status_host[2].item() == 1


def test_one_failed_block_of_a_peer_window_exchange_applies_nothing(one_rank_group, monkeypatch):
    """mopoe_comm_allreduce_adam / mopoe_comm_train_step: the exchange launch only sums;
    the Adam launch behind it applies the whole step or none of it."""
    cfg = mo.Config(**CFG)
    spec, a = make_engine(cfg)
    _, twin = make_engine(cfg)
    comm = mm.comm.XgmiComm(spec.num_floats, timeout_ms=200)
    x, eps = _batch(cfg, 256, 9), _eps(cfg, 256, 10)
    for form in ("after", "in_backward"):
        before = [t.clone() for t in (a.params, a.exp_avg, a.exp_avg_sq)]
        monkeypatch.setenv("MOPOE_TEST_XG_FAIL_SLOT", "3")   # block 3 reports a failed wait
        if form == "after":
            a.train_step(x, eps=eps, apply_adam=False)
            comm.allreduce_adam(a)
        else:
            a.train_step(x, eps=eps, apply_adam=True, comm=comm)
        torch.cuda.synchronize()
        monkeypatch.delenv("MOPOE_TEST_XG_FAIL_SLOT")
        for t0, t1 in zip(before, (a.params, a.exp_avg, a.exp_avg_sq)):
            assert torch.equal(t0, t1), form           # NO block applied its share
        assert a.adam_steps() == OrderedDict(clinical=0, rois=0)
        assert a.invalid_since() is not None
        a.recover()
    # re-armed, both forms apply the step the one-rank path applies
    a.train_step(x, eps=eps, apply_adam=True, comm=comm)
    twin.train_step(x, eps=eps)
    torch.cuda.synchronize()
    a.check_valid(sync=True)
    assert torch.equal(a.params, twin.params) and torch.equal(a.exp_avg_sq, twin.exp_avg_sq)
    assert torch.equal(a.wfrag, twin.wfrag)
    comm.close(barrier=False)


def test_data_parallel_step_rebuilds_the_weight_copies_after_its_broadcasts(one_rank_group):
    cfg = mo.Config(**CFG)
    _, a = make_engine(cfg)
    _, ref = make_engine(cfg)
    x, eps = _batch(cfg, 256, 3), _eps(cfg, 256, 4)
    a.forward(x, eps=eps)                 # (the copies are built from the first parameters)
    torch.cuda.synchronize()
    new = torch.randn(a.params.shape, generator=torch.Generator().manual_seed(8)).cuda() * 0.05
    version = a.params._version
    a.params.data.copy_(new)              # a write torch does not count, like c10d's broadcast
    ref.params.data.copy_(new)
    assert a.params._version == version
    parallel.DataParallelStep(a, exchange="allreduce")
    ref.refresh_wfrag()
    torch.cuda.synchronize()
    assert torch.equal(a.wfrag, ref.wfrag)
    a.train_step(x, eps=eps)
    ref.train_step(x, eps=eps)
    torch.cuda.synchronize()
    assert torch.equal(a.params, ref.params)


# ------------------------------------------------------------------ retry policy
def _cohort(n_subjects=1200, seed=3):
    rng = np.random.RandomState(seed)
    has_c = np.ones(n_subjects, bool)
    has_r = np.ones(n_subjects, bool)
    has_r[rng.choice(n_subjects, n_subjects // 5, replace=False)] = False
    data = {"clinical": rng.randn(int(has_c.sum()), 7), "rois": rng.randn(int(has_r.sum()), 444)}
    idx = {k: np.array([None] * n_subjects, dtype=object) for k in data}
    for name, has in (("clinical", has_c), ("rois", has_r)):
        rows = iter(range(int(has.sum())))
        for i in range(n_subjects):
            if has[i]:
                idx[name][i] = next(rows)
    return dataset.MultimodalDataset(data, idx)


class _Scripted:
    """An engine proxy for the loop: noise that belongs to the BATCH (so a retried batch
    sees the eps a twin's first attempt saw) and a hand-off time-out forced on chosen
    calls of train_step."""

    def __init__(self, eng, cfg, fail_calls=(), device_noise=False):
        self._eng, self._cfg, self._fail, self.calls = eng, cfg, set(fail_calls), 0
        self._device_noise = device_noise    # no eps injected: the kernels' own Philox draws

    def __getattr__(self, name):
        return getattr(self._eng, name)

    def train_step(self, inputs, row_index=None, **kw):
        ri = row_index if row_index is not None else inputs.row_index()   # (an IndexBatch)
        first = next(iter(ri))
        rows = ri[first]
        key = int(rows.sum().item()) * 31 + len(rows) + 7 * len(inputs)
        eps = None if self._device_noise else _eps(self._cfg, len(rows), key, list(inputs))
        fail = self.calls in self._fail
        self.calls += 1
        if fail:
            os.environ["MOPOE_TEST_HANDOFF_SPINS"] = "0"
            L.reload_knobs()
        try:
            return self._eng.train_step(inputs, row_index=row_index, eps=eps, **kw)
        finally:
            if fail:
                torch.cuda.synchronize()
                del os.environ["MOPOE_TEST_HANDOFF_SPINS"]
                L.reload_knobs()


def _exp(eng, cohort):
    model = types.SimpleNamespace(engine=eng, train=lambda: None)
    return types.SimpleNamespace(
        flags=types.SimpleNamespace(num_models=1, batch_size=256, grad_scaling=False),
        models=model, dataset_train=cohort,
        optimizers=types.SimpleNamespace(_sync=lambda: None))


def test_the_loop_retries_a_step_that_could_not_be_completed():
    cfg = mo.Config(**CFG)
    ds = _cohort()
    cohort = dataset.ResidentCohort(ds, "cuda")
    _, eng = make_engine(cfg)
    _, twin = make_engine(cfg)
    # call 1 of the epoch (its second batch: 256 rows, the fused launch) times out once
    failing, clean = _Scripted(eng, cfg, fail_calls=[1]), _Scripted(twin, cfg)
    for proxy in (failing, clean):
        np.random.seed(11)                              # the same epoch schedule
        run_epochs.train(0, 0, _exp(proxy, cohort), None)
    torch.cuda.synchronize()
    steps = len(dataset.MissingModalitySampler(ds, 256))
    assert clean.calls == steps and failing.calls > steps      # the withheld batches ran again
    eng.check_valid(sync=True)
    assert torch.equal(eng.params, twin.params)
    assert torch.equal(eng.exp_avg, twin.exp_avg) and torch.equal(eng.exp_avg_sq, twin.exp_avg_sq)
    assert eng.adam_steps() == twin.adam_steps()
    # two failures in a row: the loop gives up, parameters at the last complete step
    _, eng2 = make_engine(cfg)
    always = _Scripted(eng2, cfg, fail_calls=range(1, 64))
    np.random.seed(11)
    with pytest.raises(L.MopoeError, match="twice in a row"):
        run_epochs.train(0, 0, _exp(always, cohort), None)


def test_a_retried_step_draws_the_noise_of_its_first_attempt():
    """The device noise is keyed by (seed, step number); engine.recover() takes the step
    numbers back to the last applied step, so the replay of a withheld batch runs under its
    own number again: with the kernels' OWN noise (nothing injected) an epoch with two
    retried steps ends bit for bit where an untroubled twin of the same seed ends."""
    cfg = mo.Config(**CFG)
    ds = _cohort()
    cohort = dataset.ResidentCohort(ds, "cuda")
    _, eng = make_engine(cfg)
    _, twin = make_engine(cfg)
    assert eng.seed == twin.seed
    failing = _Scripted(eng, cfg, fail_calls=[1, 4], device_noise=True)
    clean = _Scripted(twin, cfg, device_noise=True)
    for proxy in (failing, clean):
        np.random.seed(11)
        run_epochs.train(0, 0, _exp(proxy, cohort), None)
    torch.cuda.synchronize()
    steps = len(dataset.MissingModalitySampler(ds, 256))
    assert clean.calls == steps and failing.calls >= steps + 2
    eng.check_valid(sync=True)
    assert eng.step_count() == twin.step_count() == steps      # withheld attempts do not count
    assert torch.equal(eng.params, twin.params)
    assert torch.equal(eng.exp_avg, twin.exp_avg) and torch.equal(eng.exp_avg_sq, twin.exp_avg_sq)
