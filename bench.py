#!/usr/bin/env python3
"""bench.py -- training samples/sec of the MoPoE-VAE hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], the configuration the metric is quoted
on): 2-modality joint_elbo MoPoE, input dims 7+444, latent 20, factorized
style dims [3,20], batch 256 per GPU, float32 (exact-f32 MFMA), synthetic
N(0,1) data -- a pool of 64 batches resident in HBM before the timed region
-- random-init weights.  One step = encoder/decoder forward, MoPoE fusion,
joint ELBO, full backward, (gradient all-reduce over RCCL when N > 1), Adam,
and the step's scalar log written by the kernel into pinned host memory (the
reference logs every step, run_epochs.py:184).

For N > 1 the driver launches one rank per GPU with torch.distributed.run;
ranks are data-parallel replicas (weak scaling: 256 samples per GPU per
step) that exchange the flat gradient buffer once per step: one launch per
rank over xGMI peer windows (push to every peer, rank-ordered sum, Adam --
csrc/mopoe_xgmi.inc) after a start-up check of that exchange against the
gathered inputs on this very node; if the windows cannot be set up or the
check fails, the step is RCCL all_reduce + the Adam kernel, and `config.exchange`
says which one ran (MOPOE_EXCHANGE=rccl|xgmi pins it).

Prints ONE JSON line (rank 0).  `roofline` is for the kernel with the largest
share of device time, its duration measured with HIP events on the launch
stream in a second, instrumented run of the same K steps (events perturb the
pipeline, so they stay out of the timed region that produces `value`).
`cpu_baseline` times the CPU oracle (oracle/mopoe_oracle.py, a PyTorch-CPU
restatement of the reference step) on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import mopoe_amd as mm  # noqa: E402

NAMES = ["clinical", "rois"]
DIMS = [7, 444]
STYLE = [3, 20]
LATENT = 20
BATCH = 256
POOL = int(os.environ.get("MOPOE_BENCH_POOL", "64"))   # resident batches (a diagnostic knob; 64 is what is reported)
HIDDEN = 256
# MI355X_MICROARCH.md: exact-f32 MFMA peak = vector peak; HBM3E spec
PEAK_F32_MFMA_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0


def kernel_models(spec, n, fused_adam, world=1):
    """Algorithmic flops / HBM bytes per launch of each kernel at batch n
    (DESIGN.md section 5 derives these from SURVEY.md section 8d)."""
    M = spec.num_mods
    d = sum(spec.input_dim)
    nh = sum(spec.heads_dim(m) for m in range(M))
    dz = sum(spec.input_dim[m] * spec.z_dim(m) for m in range(M))
    zd = sum(spec.z_dim(m) for m in range(M))
    P = sum(v.numel() for v in spec.param_views(
        torch.empty(spec.num_floats)).values())
    D = spec.class_dim
    S = len(spec.subset_keys)
    f = 4
    out = {}
    # h = relu(x W1^T + b1): read x and W1, write h
    out["k_linear"] = dict(
        flops=2.0 * n * HIDDEN * d,
        bytes=f * (n * d + HIDDEN * d + HIDDEN * M + n * HIDDEN * M))
    # heads, decoder, d/dz, d/dh GEMMs; reads h, x, Wh, Wd; writes heads,
    # subsets, joint, z, loc, g_xhat, g_heads, g_pre
    out["k_latent"] = dict(
        flops=2.0 * n * (2 * HIDDEN * nh + 2 * dz),
        bytes=f * (n * (2 * HIDDEN * M          # h read (+ mask re-read)
                        + d                      # x
                        + 2 * nh                 # heads, g_heads
                        + 2 * S * D + 2 * D      # subsets, joint
                        + zd + 2 * d             # z, loc, g_xhat
                        + HIDDEN * M)            # g_pre
                   + 2 * (HIDDEN * nh + dz) + nh + 3 * d))
    # G^T X for W1, Wh, Wd (+ biases) and the Adam read-modify-write
    wbytes = f * (n * (2 * HIDDEN * M + 2 * d + nh + zd) + P)
    if fused_adam:
        wbytes += f * 6 * P
    out["k_wgrad"] = dict(flops=2.0 * n * (HIDDEN * d + HIDDEN * nh + dz),
                          bytes=wbytes)
    # encoder layer + per-sample chain in one launch (small training batches): h is
    # written (for the weight gradients) and read back by the row groups all the same
    out["k_fused"] = dict(flops=out["k_linear"]["flops"] + out["k_latent"]["flops"],
                          bytes=out["k_linear"]["bytes"] + out["k_latent"]["bytes"])
    out["k_adam"] = dict(flops=0.0, bytes=f * 7 * P)
    # push to W-1 peers, read W-1 inboxes, sum written back, Adam read-modify-write
    out["k_xgmi"] = dict(flops=0.0, bytes=f * P * (8 + 2 * (world - 1)))
    out["k_finalize"] = dict(flops=0.0, bytes=0.0)
    return out


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes
    (profiles/*_pmc_traffic.json: FETCH_SIZE and WRITE_SIZE collected in
    separate runs, gfx950 correction applied); None if no summary is there."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
    if not files:
        return None
    try:
        with open(files[-1]) as f:
            return json.load(f)["kernels"][kernel]["hbm_bytes_corrected"]
    except (KeyError, ValueError, OSError):
        return None


def make_pool(device):
    g = torch.Generator().manual_seed(1234)
    pool = []
    for _ in range(POOL):
        pool.append({n: torch.randn(BATCH, d, generator=g).to(device)
                     for n, d in zip(NAMES, DIMS)})
    return pool


def cpu_baseline(seconds=10.0):
    """The oracle's train step (forward, loss, autograd backward, Adam) on the
    host cores: a bounded sample of the same workload, at two thread counts
    (these ~2,600 tiny ops do not scale with threads; the better one is `value`)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import mopoe_oracle as mo
    cfg = mo.Config(NAMES, DIMS, STYLE, class_dim=LATENT)
    g = torch.Generator().manual_seed(1234)
    pool = [{n: torch.randn(BATCH, d, generator=g) for n, d in zip(NAMES, DIMS)}
            for _ in range(8)]
    share = min(16, os.cpu_count() or 1)   # the GPU box's CPU share for one GPU
    runs = {}
    for threads in (share, 1):
        torch.set_num_threads(threads)
        params = mo.init_params(cfg, 0)
        state = mo.adam_init(params)
        noise = mo.Noise(generator=mo.noise_rng(0))
        for i in range(5):
            mo.train_step(params, cfg, pool[i % 8], noise, state)
            noise.tape.clear()
        t0 = time.perf_counter()
        steps = 0
        while time.perf_counter() - t0 < seconds / 2:
            mo.train_step(params, cfg, pool[steps % 8], noise, state)
            noise.tape.clear()
            steps += 1
        dt = time.perf_counter() - t0
        runs[threads] = (BATCH * steps / dt, steps, dt)
    best = max(runs, key=lambda k: runs[k][0])
    v, steps, dt = runs[best]
    return {"value": round(v, 1), "unit": "samples/s", "cores": best, "kind": "port",
            "sample": "%d steps of the same bs-%d joint_elbo train step (oracle/"
                      "mopoe_oracle.py: PyTorch-CPU float32 restatement of the reference "
                      "step, autograd backward, Adam) in %.1f s with %d thread(s); "
                      "host has %d logical CPUs" % (steps, BATCH, dt, best, os.cpu_count() or 0),
            "ms_per_step": round(1e3 * dt / steps, 3),
            "other_thread_counts": {str(k): round(v[0], 1) for k, v in runs.items()}}


def open_xgmi(num_floats, device, rank, world, dist):
    """Set up the peer windows and CHECK the exchange on this node before it is
    trusted with the timed region: three exchanges (both inbox parities) of
    rank-dependent data must equal, bit for bit, the rank-ordered sum of the
    inputs gathered over the process group.  Returns (comm, "") or (None, why)."""
    XgmiComm = mm.comm.XgmiComm
    try:
        comm = XgmiComm(num_floats, timeout_ms=5000)
    except mm._lib.MopoeError as e:
        return None, "peer windows unavailable: %s" % e
    ok = 1.0
    try:
        g = torch.Generator().manual_seed(4321 + rank)
        for _ in range(3):
            mine = torch.randn(num_floats, generator=g).to(device)
            parts = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(parts, mine)
            want = parts[0].clone()
            for q in parts[1:]:
                want += q
            got = comm.allreduce_(mine.clone())
            torch.cuda.synchronize()
            if not torch.equal(got, want):
                ok = 0.0
        if comm.timeouts():
            ok = 0.0
    except mm._lib.MopoeError:
        ok = 0.0
    flag = torch.tensor([ok], device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if flag.item() != 1.0:
        comm.close()
        return None, "start-up check of the xGMI exchange failed on this node"
    return comm, ""


def check_in_backward(comm, spec, batch, device, world, dist):
    """Two training steps with the exchange inside the weight-gradient launch
    (mopoe_comm_train_step) must leave the same bits as the spelled-out form:
    gradients gathered over the process group, added in rank order, Adam with 1/world."""
    ok = 1.0
    try:
        a = mm.MoPoEEngine(spec, device, seed=99)
        b = mm.MoPoEEngine(spec, device, seed=99)
        for e in (a, b):
            e.reset_parameters(torch.Generator().manual_seed(1))
        torch.cuda.synchronize()
        dist.barrier()          # ranks enter the first exchange together (bounded waits)
        for _ in range(2):
            a.train_step(batch, apply_adam=True, comm=comm)
            b.train_step(batch, apply_adam=False)
            parts = [torch.empty_like(b.grads) for _ in range(world)]
            dist.all_gather(parts, b.grads)
            total = parts[0].clone()
            for q in parts[1:]:
                total += q
            b.grads.copy_(total)
            b.adam_step(world=world)
        torch.cuda.synchronize()
        if not (torch.equal(a.params, b.params) and torch.equal(a.exp_avg_sq, b.exp_avg_sq)):
            ok = 0.0
        if comm.timeouts():
            ok = 0.0
    except mm._lib.MopoeError:
        ok = 0.0
    flag = torch.tensor([ok], device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return flag.item() == 1.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--no-log-copy", action="store_true",
                    help="leave the per-step async D2H of the scalar log out")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearse the multi-GPU step (process group, all-reduce, separate "
                         "Adam kernel) even with one rank")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with "
                     "torch.distributed.run --nproc-per-node %d"
                     % (args.gpus, args.gpus))
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    spec = mm.ModelSpec(NAMES, DIMS, STYLE, class_dim=LATENT, method="joint_elbo")
    eng = mm.MoPoEEngine(spec, device, seed=1234)
    eng.reset_parameters(torch.Generator().manual_seed(0))  # same on all ranks
    pool = make_pool(device)
    log_ring = [torch.empty(mm._lib.NUM_STATS, dtype=torch.float32).pin_memory()
                for _ in range(8)]
    fused = dist is None

    # ---- the gradient exchange of the N-rank step
    comm, exchange, why, in_backward = None, "none", "", False
    if dist is not None:
        exchange = os.environ.get("MOPOE_EXCHANGE", "auto")
        if exchange not in ("auto", "xgmi", "rccl"):
            sys.exit("MOPOE_EXCHANGE must be auto, xgmi or rccl")
        if exchange != "rccl":
            comm, why = open_xgmi(spec.num_floats, device, rank, world, dist)
            if comm is None and exchange == "xgmi":
                sys.exit("MOPOE_EXCHANGE=xgmi but: " + why)
            exchange = "xgmi" if comm is not None else "rccl"
        if comm is not None and os.environ.get("MOPOE_EXCHANGE_AFTER") is None:
            in_backward = check_in_backward(comm, spec, pool[rank % POOL], device, world, dist)
            if not in_backward:
                why = "exchange inside the weight-gradient launch failed its start-up check"

    def step(i):
        # the step's scalar log lands in a ring of pinned host buffers, written
        # by the kernel itself (no copy on the stream)
        if comm is not None and in_backward:
            return eng.train_step(pool[(i * world + rank) % POOL], apply_adam=True, comm=comm,
                                  stats_host=None if args.no_log_copy else log_ring[i % 8])[1]
        plan, ws = eng.train_step(pool[(i * world + rank) % POOL], apply_adam=fused,
                                  stats_host=None if args.no_log_copy else log_ring[i % 8])
        if comm is not None:
            comm.allreduce_adam(eng)                # one launch: push, sum, Adam
        elif not fused:
            dist.all_reduce(eng.grads)              # RCCL, one flat buffer
            eng.adam_step(world=world)
        return ws

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def timed():
        for i in range(args.warmup):
            step(i)
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            ws = step(args.warmup + i)
        barrier()
        return time.perf_counter() - t0, ws

    dt, ws = timed()
    if comm is not None:
        # the exchange must have been complete in every step, and the replicas identical
        bad = torch.tensor([comm.timeouts(), 0], device=device, dtype=torch.float64)
        mine = eng.params.double().sum()
        lo, hi = mine.clone(), mine.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        bad[1] = float(lo.item() != hi.item())
        dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        if bad[0].item() or bad[1].item():
            if os.environ.get("MOPOE_EXCHANGE") == "xgmi":
                sys.exit("xGMI exchange: %d timed-out waits, replicas differ: %s"
                         % (int(bad[0].item()), bool(bad[1].item())))
            why = "timed region over xGMI invalid (%d timeouts, replicas differ: %s); " \
                  "re-run over RCCL" % (int(bad[0].item()), bool(bad[1].item()))
            comm.close()
            comm, exchange = None, "rccl"
            eng.reset_parameters(torch.Generator().manual_seed(0))
            eng.exp_avg.zero_()
            eng.exp_avg_sq.zero_()
            dt, ws = timed()
    if dist is not None:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss = float(ws.stats[0].item())
    eng.check_valid(sync=True)     # (raises if any step of the timed region was invalid)
    if not args.no_log_copy:   # the host ring must have received the same scalar
        host = float(log_ring[(args.warmup + args.steps - 1) % 8][0])
        if host != loss:
            sys.exit("pinned-host log %r != device scalar %r" % (host, loss))
    if not (loss == loss and abs(loss) < 1e9):
        sys.exit("non-finite loss after the timed region: %r" % loss)

    out = {
        "metric": "training samples/sec (whole node), MoPoE joint_elbo, "
                  "dims 7+444, bs256",
        "value": round(BATCH * world * args.steps / dt, 1),
        "unit": "samples/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 5),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "configs[1]: 2-modality joint_elbo MoPoE-VAE "
                               "train step, input_dims 7,444, latent 20, "
                               "style 3,20, batch 256 per GPU, Adam lr 0.002",
                   "global_batch": BATCH * world,
                   "parallelism": "dp%d" % world if dist is not None else "single",
                   "exchange": {"none": "none (single GPU: Adam fused into the "
                                        "weight-gradient launch)",
                                "xgmi": ("xGMI peer windows inside the weight-gradient launch: "
                                         "every gradient block is pushed to every peer, summed "
                                         "in rank order, Adam applied" if dist is not None and
                                         comm is not None and in_backward else
                                         "one launch per rank over xGMI peer windows: push "
                                         "to every peer, rank-ordered sum, Adam"),
                                "rccl": "RCCL all_reduce of the flat buffer + Adam kernel"
                                }[exchange] + (" [%s]" % why if why else ""),
                   "host_log_every_step": not args.no_log_copy,
                   "final_loss": round(loss, 3)},
    }

    if rank == 0 and not args.no_roofline:
        # instrumented re-run of the same K steps: HIP events around every
        # launch, on the launch stream
        mm._lib.profile_enable(True)
        for i in range(args.steps):
            step(args.warmup + args.steps + i)
        torch.cuda.synchronize()
        prof = mm._lib.profile_read()
        mm._lib.profile_enable(False)
        models = kernel_models(spec, BATCH, fused or in_backward, world)
        total_ms = sum(ms for _, ms in prof.values()) or 1.0
        name = max(prof, key=lambda k: prof[k][1])
        cnt, ms = prof[name]
        raw_us = {k: v[1] / max(v[0], 1) * 1e3 for k, v in prof.items() if v[0]}
        # The two event records around a launch add device time of their own (the
        # event-timed kernels of a step sum to more than the step took in the timed
        # region, where the same kernels ran back to back).  On one GPU the step IS
        # its kernels, so that excess, spread evenly over the launches, is the event
        # overhead; it is subtracted.  The corrected figures agree with the
        # rocprofv3 --kernel-trace averages (profiles/*_kernel_stats.csv) within 4 %.
        overhead_us = 0.0
        if dist is None and fused:
            excess = sum(raw_us.values()) - 1e6 * dt / args.steps
            overhead_us = max(0.0, excess / max(len(raw_us), 1))
        avg_us = {k: v - overhead_us for k, v in raw_us.items()}
        avg_s = avg_us[name] * 1e-6
        km = models[name]
        t_mfma = km["flops"] / (PEAK_F32_MFMA_TFLOPS * 1e12)
        t_hbm = km["bytes"] / (PEAK_HBM_GBS * 1e9)
        if t_mfma >= t_hbm:
            achieved = km["flops"] / avg_s / 1e12
            roof = {"bound": "mfma", "achieved": round(achieved, 4),
                    "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 5)}
        else:
            achieved = km["bytes"] / avg_s / 1e9
            roof = {"bound": "hbm", "achieved": round(achieved, 2),
                    "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": round(achieved / PEAK_HBM_GBS, 5)}
        roof.update({"traffic": pmc_traffic(name), "kernel": name,
                     "avg_us": round(avg_s * 1e6, 3),
                     "share_of_device_time": round(ms / total_ms, 3),
                     "kernels_avg_us": {k: round(v, 3) for k, v in avg_us.items()},
                     "kernels_avg_us_with_event_overhead":
                         {k: round(v, 3) for k, v in raw_us.items()},
                     "event_overhead_us_per_launch": round(overhead_us, 3),
                     "algorithmic_flops": km["flops"],
                     "algorithmic_bytes": km["bytes"]})
        out["roofline"] = roof
    elif dist is not None and not args.no_roofline:
        for i in range(args.steps):   # keep the collectives matched
            step(args.warmup + args.steps + i)
        torch.cuda.synchronize()

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    if comm is not None:
        comm.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
