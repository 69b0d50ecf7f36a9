"""One rank of tests/test_hip_xgmi.py (started as a subprocess: RANK, WORLD_SIZE,
MASTER_ADDR, MASTER_PORT in the environment; LOCAL_RANK picks the GPU, and the
ranks may all sit on ONE GPU -- peer windows work between processes of the
same device too, which is how a one-GPU box rehearses the node's exchange).
The process group is gloo: it only carries the IPC handles and the checks."""
import os
import sys
from collections import OrderedDict
from importlib import import_module

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import mopoe_amd as mm  # noqa: E402

comm_mod = import_module("2022_cambroise_interpret_multivae_amd.comm")
parallel = import_module("2022_cambroise_interpret_multivae_amd.parallel")


def gather_cpu(t):
    """Every rank's tensor, on the host, in rank order."""
    out = [torch.empty_like(t, device="cpu") for _ in range(dist.get_world_size())]
    dist.all_gather(out, t.detach().cpu())
    return out


def ordered_sum(parts):
    s = parts[0].clone()
    for p in parts[1:]:
        s += p                      # ((v0 + v1) + v2) + ... : the kernel's order
    return s


def main():
    rank = int(os.environ["RANK"])
    world = int(os.environ["WORLD_SIZE"])
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    names, style = ["clinical", "rois"], [3, 20]
    dims = [int(d) for d in os.environ.get("XGMI_DIMS", "7,444").split(",")]
    spec = mm.ModelSpec(names, dims, style, class_dim=20, method="joint_elbo")
    eng = mm.MoPoEEngine(spec, dev, seed=77)
    eng.reset_parameters(torch.Generator().manual_seed(0))
    P = spec.num_floats
    comm = comm_mod.XgmiComm(P, timeout_ms=int(os.environ.get("XGMI_TIMEOUT_MS", "3000")))

    # 1. plain exchange, back to back (both inbox parities, no host sync between)
    g = torch.Generator().manual_seed(100 + rank)
    rounds = [torch.randn(P, generator=g) for _ in range(7)]
    dev_rounds = [r.to(dev) for r in rounds]
    for t in dev_rounds:
        comm.allreduce_(t)
    torch.cuda.synchronize()
    if os.environ.get("XGMI_TIME"):
        import time
        scratch = torch.zeros(P, device=dev)
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(500):
            comm.allreduce_(scratch)
        torch.cuda.synchronize()
        print("rank %d: %.2f us per exchange of %d floats, %d ranks on one GPU" % (
            rank, (time.perf_counter() - t0) / 500 * 1e6, P, world), flush=True)
    for i, (t, r) in enumerate(zip(dev_rounds, rounds)):
        want = ordered_sum(gather_cpu(r))
        if not torch.equal(t.cpu(), want):
            raise SystemExit("rank %d: round %d of the plain exchange differs from the "
                             "rank-ordered sum (max %g)" % (
                                 rank, i, (t.cpu() - want).abs().max().item()))

    # 2. the data-parallel step: exchange + Adam in one launch vs the spelled-out form
    #    (gather, rank-ordered sum on the host, mopoe_adam_step with 1/world)
    ref = mm.MoPoEEngine(spec, dev, seed=77)
    ref.reset_parameters(torch.Generator().manual_seed(0))
    gx = torch.Generator().manual_seed(500 + rank)
    prev = None
    for it in range(4):
        n = 64
        batch = OrderedDict((k, torch.randn(n, d, generator=gx).to(dev))
                            for k, d in zip(names, dims))
        if it == 2:                              # a batch without `clinical`
            batch.pop("clinical")
        eps = None
        ref.train_step(batch, eps=eps, apply_adam=False)
        # same Philox seed and step number -> the same noise as `eng` below
        total = ordered_sum(gather_cpu(ref.grads))
        ref.grads.copy_(total.to(dev))
        ref.adam_step(world=world)
        eng.train_step(batch, eps=eps, apply_adam=False)
        local = gather_cpu(eng.grads)
        if not torch.equal(eng.grads.cpu(), local[rank]) or \
                not torch.equal(ordered_sum(local), total):
            raise SystemExit("rank %d step %d: the two engines' local gradients differ"
                             % (rank, it))
        comm.allreduce_adam(eng)
        torch.cuda.synchronize()
        if not torch.equal(eng.grads.cpu(), total):
            why = []
            for r in range(world):
                if prev is not None:
                    alt = list(local)
                    alt[r] = prev[r]
                    if torch.equal(eng.grads.cpu(), ordered_sum(alt)):
                        why.append("rank %d's PREVIOUS gradients were summed" % r)
            raise SystemExit("rank %d step %d: exchanged sum is wrong (%s; %d timeouts)"
                             % (rank, it, "; ".join(why) or "no single stale rank explains it",
                                comm.timeouts()))
        prev = local
        for name, a, b in (("grads", eng.grads, ref.grads), ("params", eng.params, ref.params),
                           ("exp_avg", eng.exp_avg, ref.exp_avg),
                           ("exp_avg_sq", eng.exp_avg_sq, ref.exp_avg_sq)):
            if not torch.equal(a, b):
                d = (a - b).abs()
                raise SystemExit("rank %d step %d: %s differs from all-reduce + k_adam "
                                 "(max %g, %d elements)" % (rank, it, name, d.max().item(),
                                                            int((d > 0).sum())))
        everyone = gather_cpu(eng.params)
        if not all(torch.equal(everyone[0], e) for e in everyone[1:]):
            raise SystemExit("rank %d step %d: replicas drifted apart" % (rank, it))

    # 3. twenty steps through the DataParallelStep wrapper, then the same twenty spelled
    #    out from the same state.  The ranks go in lock-step here (a barrier per step)
    #    because they share ONE GPU: a polling exchange block of a rank that is a step
    #    ahead sits on a CU that a 1024-thread workgroup of the rank it waits for needs
    #    -- a circular wait over compute units that cannot arise on the node, where each
    #    rank has its own GPU.  (Ranks running a step apart are covered by part 1.)
    state = [t.clone() for t in (eng.params, eng.exp_avg, eng.exp_avg_sq, eng.counters)]
    batches = [OrderedDict((k, torch.randn(64, d, generator=gx).to(dev))
                           for k, d in zip(names, dims)) for _ in range(20)]
    step = parallel.DataParallelStep(eng, comm=comm, exchange="xgmi")
    for b in batches:
        step(b)
        torch.cuda.synchronize()
        dist.barrier()
    for dst, src in zip((ref.params, ref.exp_avg, ref.exp_avg_sq, ref.counters), state):
        dst.copy_(src)
    for b in batches:
        ref.train_step(b, apply_adam=False)
        ref.grads.copy_(ordered_sum(gather_cpu(ref.grads)).to(dev))
        ref.adam_step(world=world)
    torch.cuda.synchronize()
    eng.check_valid(sync=True)
    ref.check_valid(sync=True)
    if not torch.equal(eng.params, ref.params):
        d = (eng.params - ref.params).abs()
        raise SystemExit("rank %d: 20 back-to-back steps differ from the spelled-out form "
                         "(max %g, %d elements; %d timeouts)" % (
                             rank, d.max().item(), int((d > 0).sum()), comm.timeouts()))

    # 4. the exchange inside the weight-gradient launch (mopoe_comm_train_step) against
    #    the spelled-out form, from the same state: eight steps, the last one on a batch
    #    without `clinical` (fewer blocks in the launch, same on all ranks)
    eight = batches[:7] + [OrderedDict((k, v) for k, v in batches[7].items()
                                       if k != "clinical")]
    # one batch above 512 rows: the 8-wave form of the weight-gradient launch (half of its
    # waves own no output and only keep the exchange's barriers company)
    eight[3] = OrderedDict((k, torch.randn(600, d, generator=gx).to(dev))
                           for k, d in zip(names, dims))

    def restore(e):
        for dst, src in zip((e.params, e.exp_avg, e.exp_avg_sq, e.counters), state):
            dst.copy_(src)

    if os.environ.get("XGMI_IN_BACKWARD", "1") == "0":
        eight = []
    restore(eng)
    step = parallel.DataParallelStep(eng, comm=comm, exchange="xgmi_in_backward")
    for b in eight:
        step(b)
        torch.cuda.synchronize()
        dist.barrier()
    restore(ref)
    for b in eight:
        ref.train_step(b, apply_adam=False)
        ref.grads.copy_(ordered_sum(gather_cpu(ref.grads)).to(dev))
        ref.adam_step(world=world)
    torch.cuda.synchronize()
    pairs = [("params", eng.params, ref.params), ("exp_avg", eng.exp_avg, ref.exp_avg),
             ("exp_avg_sq", eng.exp_avg_sq, ref.exp_avg_sq)]
    ga, gr = spec.param_views(eng.grads), spec.param_views(ref.grads)
    pairs += [("grad of " + k, ga[k], gr[k]) for k in ga if "rois" in k]
    for name, a, c in pairs:
        if not torch.equal(a, c):
            d = (a - c).abs()
            raise SystemExit("rank %d: exchange inside the weight-gradient launch: %s differs "
                             "from all-reduce + k_adam (max %g, %d elements; %d timeouts)" % (
                                 rank, name, d.max().item(), int((d > 0).sum()),
                                 comm.timeouts()))
    if comm.timeouts() != 0:
        raise SystemExit("rank %d: %d waits timed out" % (rank, comm.timeouts()))
    comm.close()
    dist.barrier()
    dist.destroy_process_group()
    print("rank %d ok" % rank, flush=True)


if __name__ == "__main__":
    main()
