#!/usr/bin/env python3
"""bench.py -- training samples/sec of the MoPoE-VAE hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload of `value` (BASELINE.json configs[1], the configuration the metric is quoted
on): 2-modality joint_elbo MoPoE, input dims 7+444, latent 20, factorized style dims
[3,20], batch 256 per GPU, float32 (exact-f32 MFMA), synthetic N(0,1) data -- a pool of
64 batches resident in HBM before the timed region -- random-init weights.  One step =
encoder/decoder forward, MoPoE fusion, joint ELBO, full backward, (gradient exchange
when N > 1), Adam, and the step's scalar log written by the kernel into pinned host
memory (the reference logs every step, run_epochs.py:184).  W untimed steps, then
EXACTLY K timed steps between barrier + synchronize pairs, max over ranks -- as the first
GPU work of the process (`cold_start`: at --steps 20 --warmup 5 the whole timed region is
under a millisecond of a chip that was idle, and reads ~25 % slow), and again after --settle
(3000) untimed steps, which is `value`: what training runs at.  For K <= 200 that second
region is timed five times back to back and `value` is the median (a 0.6 ms region between
two synchronisations read 30.7 .. 42.7 us per step over four runs on one box); all samples
are in the line.

For N > 1 the driver launches one rank per GPU with torch.distributed.run; ranks are
data-parallel replicas (weak scaling: 256 samples per GPU per step) that exchange the
flat gradient buffer once per step with RCCL's all-reduce (+ the Adam kernel, which
also checks that every rank's batch held the same modalities and that every rank
completed its backward) -- ONE host call per step: mopoe_rccl_train_step enqueues the
backward, ncclAllReduce and the Adam launch over the library's own RCCL communicator.
MOPOE_EXCHANGE=c10d spells the same step out (train_step, torch.distributed.all_reduce,
adam_step: three host calls); MOPOE_EXCHANGE=xgmi opts into the peer-window exchange of
csrc/mopoe_xgmi.inc (start-up checked against the gathered inputs on the node it runs on;
it has never run on more than one GPU).

Prints ONE JSON line (rank 0).  Besides the contract's keys:
  roofline       the kernel with the largest share of device time, its duration measured
                 with HIP events on the launch stream in a second, instrumented run of
                 the same K steps (events perturb the pipeline, so they stay out of the
                 timed region that produces `value`)
  cpu_baseline   the CPU oracle (oracle/mopoe_oracle.py, a PyTorch-CPU restatement of the
                 reference step) on this box's host cores: all cores and 1 thread
  cold_start     the first of the two timings (see above)
  long_run       the --settle steps between the two, timed as one region
  other_configs  BASELINE.json configs[2] (method poe, batch 1024) and configs[4]
                 (4 modalities, 15 subsets, batch 512) on one GPU: samples/s + roofline
  regime_n65536  the kernels at 65,536 rows, where the batch term dominates
  eager_rocm_baseline   the oracle with its tensors on the GPU (PyTorch-ROCm eager):
                 the "HIP kernels vs PyTorch-ROCm eager" leg of configs[1]
(the last four on rank 0 of a 1-GPU run only; --quick leaves them out).
"""
import argparse
import json
import os
import sys
import time

CONFIGS = {
    # BASELINE.json configs[0] / [1] / [3]
    "C1": dict(names=["clinical", "rois"], dims=[7, 444], style=[3, 20], method="joint_elbo",
               batch=256, label="configs[1]: 2-modality joint_elbo MoPoE-VAE train step, "
                                "input_dims 7,444, latent 20, style 3,20, batch 256 per GPU, "
                                "Adam lr 0.002"),
    # configs[2]
    "C3": dict(names=["clinical", "rois"], dims=[7, 444], style=[3, 20], method="poe",
               batch=1024, label="configs[2]: method poe (joint + unimodal ELBOs), 2 modalities, "
                                 "batch 1024"),
    # configs[4] (style dims [3,3,3,3]: experiment.py:133-136 repeats style_dim[0])
    "C5": dict(names=["clinical", "rois", "snps", "tracts"], dims=[7, 444, 128, 64],
               style=[3, 3, 3, 3], method="joint_elbo", batch=512,
               label="configs[4]: 4 modalities (7,444,128,64), joint_elbo over the 15-subset "
                     "powerset, batch 512 on one GPU"),
    # SURVEY section 8d's second regime: configs[1]'s model where the batch term dominates
    # (profiling only: tools/profile_round.sh <tag> --config N64K)
    "N64K": dict(names=["clinical", "rois"], dims=[7, 444], style=[3, 20], method="joint_elbo",
                 batch=65536, label="configs[1]'s model at 65,536 rows per step (the MFMA / "
                                    "bandwidth regime; not a BASELINE configuration)"),
}
LATENT = 20


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--settle", type=int, default=3000,
                    help="untimed steps between the cold-start timing and the timing that "
                         "produces `value` (reported as config.settle_steps)")
    ap.add_argument("--no-log-copy", action="store_true",
                    help="leave the per-step scalar log into pinned host memory out")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--quick", action="store_true",
                    help="headline + roofline + cpu_baseline only (no other "
                         "configs / regime / eager legs)")
    ap.add_argument("--config", choices=sorted(CONFIGS), default="C1",
                    help="the configuration the main loop runs (profiling: tools/"
                         "profile_round.sh); the metric is quoted on C1 = configs[1]")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="rows per step over ALL ranks (each rank takes global / N; strong "
                         "scaling).  BASELINE configs[4] as SURVEY 8d states it: --config C5 "
                         "--global-batch 512 --gpus 8 = 64 rows per rank.  Default: the "
                         "configuration's batch PER rank (weak scaling)")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearse the multi-GPU step (rank spawn, process group, RCCL "
                         "all-reduce, separate Adam kernel) even with one rank")
    ap.add_argument("--launch-dry-run", action="store_true",
                    help="print the N child environments / command lines the launcher would "
                         "start, as one JSON line, and start nothing")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------
# The launcher: `python3 bench.py --gpus N` (no torchrun around it) starts its own N ranks.
# It runs BEFORE torch, the package or anything else that could touch the GPU is imported: the
# parent only spawns CHILD processes (never an exec, never a re-launch of a process that has
# initialised HIP), waits for them, forwards rank 0's single JSON line and returns non-zero if
# any rank did.  Under torch.distributed.run (WORLD_SIZE set) this is skipped: the process IS a rank.
def rank_environments(n, port=None):
    """The N child environments (only the variables the launcher sets)."""
    if port is None:
        import socket
        with socket.socket() as s:        # a free port of the loopback interface
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    return [{"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
             "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
             # dmabuf IPC (the host driver supports nothing else: RCCL / peer windows need it)
             "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
             "MOPOE_BENCH_SPAWNED": "1"} for r in range(n)]


def launcher(args, argv, cmd=None):
    """`cmd`: the child command line (tests: a stand-in for a rank); default: this file."""
    import subprocess
    n = args.gpus
    if n < 1:
        sys.exit("--gpus must be >= 1")
    child_argv = [a for a in argv if a != "--launch-dry-run"]
    envs = rank_environments(n, port=29400 if args.launch_dry_run else None)
    if cmd is None:
        cmd = [sys.executable, os.path.abspath(__file__)] + child_argv
    if args.launch_dry_run:
        print(json.dumps({"launch": "child processes (subprocess.Popen), one per GPU", "n_gpus": n,
                          "cmd": cmd, "ranks": envs}), flush=True)
        return 0
    procs = []
    for r, e in enumerate(envs):
        # rank 0's stdout carries the JSON line; the other ranks' stdout goes to stderr
        procs.append(subprocess.Popen(cmd, env=dict(os.environ, **e),
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    line = b""
    for raw in procs[0].stdout:           # (until rank 0 closes its stdout: it has ended)
        if raw.strip():
            line = raw
    rcs = []
    deadline = time.time() + 120.0        # the others end with rank 0 (last barrier) or are stopped
    for p in procs:
        try:
            rcs.append(p.wait(timeout=max(1.0, deadline - time.time())))
        except subprocess.TimeoutExpired:
            p.kill()                       # (this exact child: never a pattern)
            rcs.append(p.wait())
    if any(rcs):
        sys.stderr.write("bench.py: rank exit codes %s\n" % rcs)
        return next(rc for rc in rcs if rc) or 1
    try:
        json.loads(line)
    except ValueError:
        sys.stderr.write("bench.py: rank 0 printed no JSON line\n")
        return 1
    sys.stdout.write(line.decode())
    sys.stdout.flush()
    return 0


def wants_launcher(args):
    """This process is the launcher (not a rank): nobody gave it a rank environment, and more
    than one rank -- or the rehearsal of the rank-spawn path -- was asked for."""
    return "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.force_dist or args.launch_dry_run)


if __name__ == "__main__" and wants_launcher(parse_args()):
    sys.exit(launcher(parse_args(), sys.argv[1:]))     # (before torch / the package are imported)

# The package makes the host wait for the GPU by polling (HSA_ENABLE_INTERRUPT=0;
# MOPOE_HOST_WAIT=interrupt keeps ROCm's default) -- its policy, 2022_cambroise_interpret_multivae_amd/
# _lib.py, so that the benchmark and a training run through run_epochs.train wait the same way.  It
# is applied here as well, ahead of `import torch` (the runtime reads the variable once, when it
# starts).  What it buys: the first synchronisations of a process (cold_start 34-36 us per step
# against 49-70 with interrupts, same box); the 20-step figure itself is noise-limited either
# way (30.7-42.7 polling, 31.7-32.1 interrupts over six runs).  Reported as config.host_wait.
if os.environ.get("MOPOE_HOST_WAIT", "poll") != "interrupt":
    os.environ.setdefault("HSA_ENABLE_INTERRUPT", "0")
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import mopoe_amd as mm  # noqa: E402

HIDDEN = 256
# MI355X_MICROARCH.md: exact-f32 MFMA peak = vector peak; HBM3E spec
PEAK_F32_MFMA_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0
POOL = int(os.environ.get("MOPOE_BENCH_POOL", "64"))   # resident batches



def make_spec(c):
    return mm.ModelSpec(c["names"], c["dims"], c["style"], class_dim=LATENT, method=c["method"])


def kernel_models(spec, n, fused_adam, world=1):
    """Algorithmic flops / HBM bytes per launch of each kernel at batch n
    (DESIGN.md section 5 derives these from SURVEY.md section 8d).  Method poe decodes
    every modality twice (joint latent + its unimodal posterior)."""
    M = spec.num_mods
    passes = 2 if spec.method == "poe" and spec.poe_unimodal_elbos else 1
    d = sum(spec.input_dim)
    nh = sum(spec.heads_dim(m) for m in range(M))
    dz = sum(spec.input_dim[m] * spec.z_dim(m) for m in range(M))
    zd = sum(spec.z_dim(m) for m in range(M))
    P = sum(v.numel() for v in spec.param_views(torch.empty(spec.num_floats)).values())
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
        flops=2.0 * n * (2 * HIDDEN * nh + 2 * passes * dz),
        bytes=f * (n * (2 * HIDDEN * M          # h read (+ mask re-read)
                        + d                      # x
                        + 2 * nh                 # heads, g_heads
                        + 2 * S * D + 2 * D      # subsets, joint
                        + passes * (zd + 2 * d)  # z, loc, g_xhat
                        + HIDDEN * M)            # g_pre
                   + 2 * (HIDDEN * nh + dz) + nh + 3 * d))
    # G^T X for W1, Wh, Wd (+ biases) and the Adam read-modify-write
    wbytes = f * (n * (2 * HIDDEN * M + d + nh + passes * (d + zd)) + P)
    if fused_adam:
        wbytes += f * 6 * P
    out["k_wgrad"] = dict(flops=2.0 * n * (HIDDEN * d + HIDDEN * nh + passes * dz), bytes=wbytes)
    # encoder layer + per-sample chain in one launch (small training batches): h is
    # written (for the weight gradients) and read back by the row groups all the same
    out["k_fused"] = dict(flops=out["k_linear"]["flops"] + out["k_latent"]["flops"],
                          bytes=out["k_linear"]["bytes"] + out["k_latent"]["bytes"])
    out["k_adam"] = dict(flops=0.0, bytes=f * 7 * P)
    # push to W-1 peers, read W-1 inboxes, sum written back, Adam read-modify-write
    out["k_xgmi"] = dict(flops=0.0, bytes=f * P * (2 + 2 * (world - 1)))
    # ring all-reduce: every rank sends and receives 2 (W-1)/W of the buffer
    out["rccl_allreduce"] = dict(flops=0.0, bytes=f * P * 2.0 * 2.0 * (world - 1) / max(world, 1))
    out["k_finalize"] = dict(flops=0.0, bytes=0.0)
    return out


def pmc_traffic(kernel, cfg="C1"):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes
    (profiles/*_pmc_traffic[_<config>].json: FETCH_SIZE and WRITE_SIZE collected in separate
    runs of `bench.py --config <config>`, gfx950 correction applied); None if no summary of
    that configuration is there."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic_%s.json" % cfg)))
    if not files and cfg == "C1":
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
    if not files:
        return None
    try:
        with open(files[-1]) as f:
            return json.load(f)["kernels"][kernel]["hbm_bytes_corrected"]
    except (KeyError, ValueError, OSError):
        return None


def pmc_insts(kernel, cfg="C1"):
    """The SQ instruction-issue counters of `kernel` from the newest committed pass
    (profiles/*_pmc_insts_<config>.json, tools/pmc_insts.py: rocprofv3 --pmc over `bench.py
    --config <config>`), reduced to the fractions the roofline object carries; None without one.
      mfma_busy_frac   SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES): the share of the
                       SIMD-cycles of the CUs that held a workgroup, while they held one, in which
                       the MFMA pipe was busy (x share of CUs busy = chip-level utilisation)
      wave cycles      parked on s_waitcnt / barrier (wait_any), issue-stalled (wait_inst_any),
                       issuing (active_inst_any; of which vector ALU: active_inst_valu)"""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_insts_%s.json" % cfg)))
    if not files:
        return None
    try:
        with open(files[-1]) as f:
            ks = json.load(f)["kernels"]
        key = next((k for k in ks if k == kernel or k.startswith(kernel + "<")), None)
        d, p = ks[key]["derived"], ks[key]["per_launch"]
        out = {"source": os.path.basename(files[-1]), "kernel": key,
               "mfma_busy_frac": d.get("mfma_busy_frac_of_busy_cu_simd_cycles"),
               "mfma_busy_frac_of_chip": d.get("mfma_busy_frac_of_chip"),
               "wait_any_frac": d.get("wait_any_frac_of_wave_cycles"),
               "wait_inst_any_frac": d.get("wait_inst_any_frac_of_wave_cycles"),
               "active_inst_any_frac": d.get("active_inst_any_frac_of_wave_cycles"),
               "active_inst_valu_frac": d.get("active_inst_valu_frac_of_wave_cycles"),
               "active_inst_scalar_frac": d.get("active_inst_scalar_frac_of_wave_cycles"),
               "insts_per_wave": {k[:-9]: v for k, v in d.items() if k.endswith("_per_wave") and k.startswith("insts_")},
               "waves": p.get("SQ_WAVES")}
        return out
    except (KeyError, ValueError, OSError, TypeError):
        return None


def make_pool(c, device, count=POOL, seed=1234):
    g = torch.Generator().manual_seed(seed)
    return [{n: torch.randn(c["batch"], d, generator=g).to(device)
             for n, d in zip(c["names"], c["dims"])} for _ in range(count)]


def roofline_of(spec, n, prof, dt_per_step, fused_adam, world, traffic="C1"):
    """The roofline object for the kernel with the largest share of device time."""
    models = kernel_models(spec, n, fused_adam, world)
    total_ms = sum(ms for _, ms in prof.values()) or 1.0
    name = max(prof, key=lambda k: prof[k][1])
    cnt, ms = prof[name]
    raw_us = {k: v[1] / max(v[0], 1) * 1e3 for k, v in prof.items() if v[0]}
    # The two event records around a launch add device time of their own (the event-timed
    # kernels of a step sum to more than the step took in the timed region, where the same
    # kernels ran back to back).  On one GPU the step IS its kernels, so that excess,
    # spread evenly over the launches, is the event overhead; it is subtracted.  The
    # corrected figures agree with the rocprofv3 --kernel-trace averages
    # (profiles/*_kernel_stats.csv) within 4 %.
    overhead_us = 0.0
    if dt_per_step is not None:
        excess = sum(raw_us.values()) - 1e6 * dt_per_step
        overhead_us = max(0.0, excess / max(len(raw_us), 1))
    avg_us = {k: v - overhead_us for k, v in raw_us.items()}
    avg_s = avg_us[name] * 1e-6
    km = models[name]
    if name == "k_fused" and prof.get("k_linear", (0, 0))[0]:
        # (row groups only: the encoder layer ran as a launch of its own in front of them --
        #  four-row groups beyond 512 rows -- so its work is not this launch's)
        km = models["k_latent"]
    t_mfma = km["flops"] / (PEAK_F32_MFMA_TFLOPS * 1e12)
    t_hbm = km["bytes"] / (PEAK_HBM_GBS * 1e9)
    if t_mfma >= t_hbm:
        achieved = km["flops"] / avg_s / 1e12
        roof = {"bound": "mfma", "achieved": round(achieved, 4), "peak": PEAK_F32_MFMA_TFLOPS,
                "unit": "TFLOP/s", "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 5)}
    else:
        achieved = km["bytes"] / avg_s / 1e9
        roof = {"bound": "hbm", "achieved": round(achieved, 2), "peak": PEAK_HBM_GBS,
                "unit": "GB/s", "frac": round(achieved / PEAK_HBM_GBS, 5)}
    insts = pmc_insts(name, traffic) if traffic else None
    roof.update({"traffic": pmc_traffic(name, traffic) if traffic else None, "kernel": name,
                 "mfma_busy_frac": insts["mfma_busy_frac"] if insts else None,
                 "issue_counters": insts,
                 "avg_us": round(avg_s * 1e6, 3),
                 "share_of_device_time": round(ms / total_ms, 3),
                 "kernels_avg_us": {k: round(v, 3) for k, v in avg_us.items()},
                 "kernels_avg_us_with_event_overhead": {k: round(v, 3) for k, v in raw_us.items()},
                 "event_overhead_us_per_launch": round(overhead_us, 3),
                 "algorithmic_flops": km["flops"], "algorithmic_bytes": km["bytes"]})
    return roof


def time_single_gpu(c, device, steps, warmup, log=True):
    """(seconds for `steps` steps, engine, step fn) of config `c` on one GPU."""
    spec = make_spec(c)
    eng = mm.MoPoEEngine(spec, device, seed=1234)
    eng.reset_parameters(torch.Generator().manual_seed(0))
    pool = make_pool(c, device, count=min(POOL, max(8, (1 << 28) // (4 * c["batch"] * sum(c["dims"])))))
    ring = [torch.empty(mm._lib.NUM_STATS, dtype=torch.float32).pin_memory() for _ in range(8)]

    def step(i):
        return eng.train_step(pool[i % len(pool)], apply_adam=True,
                              stats_host=ring[i % 8] if log else None)[1]

    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    eng.check_valid(sync=True)
    return dt, eng, step, spec


def profile_steps(step, first, steps):
    mm._lib.profile_enable(True)
    for i in range(steps):
        step(first + i)
    torch.cuda.synchronize()
    prof = mm._lib.profile_read()
    mm._lib.profile_enable(False)
    return prof


def other_config(key, device, steps=1000, warmup=300):
    c = CONFIGS[key]
    dt, eng, step, spec = time_single_gpu(c, device, steps, warmup)
    prof = profile_steps(step, warmup + steps, steps)
    loss = float(eng._ws[next(iter(eng._ws))].stats[0])
    return {"workload": c["label"], "value": round(c["batch"] * steps / dt, 1),
            "unit": "samples/s", "ms_per_step": round(1e3 * dt / steps, 5), "steps": steps,
            "warmup": warmup, "dtype": "f32", "final_loss": round(loss, 3),
            "roofline": roofline_of(spec, c["batch"], prof, dt / steps, True, 1, traffic=key)}


def regime_point(device, n=65536, operands="f32"):
    """The kernels of configs[1]'s model at 65,536 rows (the
    bandwidth / MFMA regime, where the batch term dominates the per-step terms).
    operands="bf16": the opt-in that BASELINE configs[1] names (the encoder layer's GEMM
    on bfloat16 roundings, float32 sums) -- reported NEXT to the float32 figures, never
    instead of them."""
    c = dict(CONFIGS["C1"], batch=n)
    spec = mm.ModelSpec(c["names"], c["dims"], c["style"], class_dim=LATENT, method=c["method"],
                        gemm_operands=operands)
    eng = mm.MoPoEEngine(spec, device, seed=1)
    eng.reset_parameters(torch.Generator().manual_seed(0))
    pool = make_pool(c, device, count=2)
    for i in range(3):
        eng.train_step(pool[i % 2])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps = 10
    for i in range(steps):
        eng.train_step(pool[i % 2])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    mm._lib.profile_enable(True)
    for i in range(steps):
        eng.train_step(pool[i % 2])
    torch.cuda.synchronize()
    prof = mm._lib.profile_read()
    mm._lib.profile_enable(False)
    models = kernel_models(spec, n, True)
    out = {"rows": n, "encoder_layer_operands": operands, "ms_per_step": round(1e3 * dt, 4),
           "samples_per_s": round(n / dt, 1), "kernels": {}}
    for k, (cnt, ms) in prof.items():
        if not cnt:
            continue
        us = ms / cnt * 1e3
        km = models[k]
        out["kernels"][k] = {
            "avg_us": round(us, 2),
            "tflops": round(km["flops"] / (us * 1e-6) / 1e12, 2),
            "frac_f32_mfma_peak": round(km["flops"] / (us * 1e-6) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
            "algorithmic_gbs": round(km["bytes"] / (us * 1e-6) / 1e9, 1)}
    return out


def loop_figure(device, subjects, batch=256, epochs=20):
    """The drop-in loop (SURVEY.md 8d's second number): run_epochs.train of the mirror
    package over a ResidentCohort -- a synthetic cohort of `subjects` subjects, 20 % of them
    lacking one block, blocks resident in HBM, the reference's MissingModalitySampler draws
    (its legacy np.random stream, restated in C and drawn one epoch ahead by a helper thread),
    index batches gathered by the kernels, one fused step per batch, the retry policy's
    per-step look.  us per step next to the bare engine loop of `value`."""
    import types
    import numpy as np
    from importlib import import_module
    P = "2022_cambroise_interpret_multivae_amd."
    ds_mod = import_module(P + "multimodal_cohort.dataset")
    run_epochs = import_module(P + "run_epochs")
    c = CONFIGS["C1"]
    rng = np.random.RandomState(0)
    lacks, which = rng.rand(subjects) < 0.2, rng.rand(subjects) < 0.5
    has = {"clinical": ~(lacks & which), "rois": ~(lacks & ~which)}
    data, idx = {}, {}
    for mod, dim in zip(c["names"], c["dims"]):
        rows = np.flatnonzero(has[mod])
        data[mod] = rng.randn(len(rows), dim)
        col = np.empty(subjects, dtype=object)
        col[:] = None
        for k, subj in enumerate(rows):
            col[subj] = k
        idx[mod] = col
    ds = ds_mod.MultimodalDataset(data, idx)
    cohort = ds_mod.ResidentCohort(ds, device, scalers=ds_mod.fit_scalers(ds))
    eng = mm.MoPoEEngine(make_spec(c), device, seed=7)
    eng.reset_parameters(torch.Generator().manual_seed(0))
    exp = types.SimpleNamespace(
        flags=types.SimpleNamespace(num_models=1, batch_size=batch, grad_scaling=False),
        models=types.SimpleNamespace(engine=eng, train=lambda: None), dataset_train=cohort,
        optimizers=types.SimpleNamespace(_sync=lambda: None))
    steps = len(ds_mod.MissingModalitySampler(ds, batch))
    np.random.seed(1)
    for _ in range(3):
        run_epochs.train(0, 0, exp, None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(epochs):
        run_epochs.train(0, 0, exp, None)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    eng.check_valid(sync=True)
    return {"subjects": subjects, "batch": batch, "steps_per_epoch": steps, "epochs": epochs,
            "us_per_step": round(1e6 * dt / (epochs * steps), 2),
            "samples_per_s": round(epochs * subjects / dt, 1)}


def oracle_setup(c):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import mopoe_oracle as mo
    cfg = mo.Config(c["names"], c["dims"], c["style"], class_dim=LATENT, method=c["method"])
    return mo, cfg


def cpu_baseline(seconds=18.0):
    """The oracle's train step (forward, loss, autograd backward, Adam) on the host cores:
    a bounded sample of configs[1]'s workload at 1 thread and at the 16 threads that are a
    one-GPU job's share of the box (or all of them on a smaller host).  There is no
    all-cores leg: the step is ~2,600 tiny torch ops that do not scale with threads, and on
    the 256-thread GPU host the all-cores setting took 16.3 s for ONE step (BENCH_r02.json:
    oversubscription, not a baseline).  The best setting is `value`, `cores` its threads."""
    c = CONFIGS["C1"]
    mo, cfg = oracle_setup(c)
    g = torch.Generator().manual_seed(1234)
    pool = [{n: torch.randn(c["batch"], d, generator=g) for n, d in zip(c["names"], c["dims"])}
            for _ in range(8)]
    allc = os.cpu_count() or 1
    try:    # the cores this process may actually run on
        allc = min(allc, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    runs = {}
    before = torch.get_num_threads()
    settings = sorted({1, min(16, allc)})
    for threads in settings:
        torch.set_num_threads(threads)
        params = mo.init_params(cfg, 0)
        state = mo.adam_init(params)
        noise = mo.Noise(generator=mo.noise_rng(0))
        t0 = time.perf_counter()
        mo.train_step(params, cfg, pool[0], noise, state)      # warm-up (+ a probe)
        noise.tape.clear()
        slow = time.perf_counter() - t0 > 0.5
        for i in range(0 if slow else 4):
            mo.train_step(params, cfg, pool[i % 8], noise, state)
            noise.tape.clear()
        budget = seconds / len(settings)
        t0 = time.perf_counter()
        steps = 0
        while time.perf_counter() - t0 < budget and not (slow and steps >= 2):
            mo.train_step(params, cfg, pool[steps % 8], noise, state)
            noise.tape.clear()
            steps += 1
        dt = time.perf_counter() - t0
        runs[threads] = (c["batch"] * steps / dt, steps, dt)
    torch.set_num_threads(before)
    best = max(runs, key=lambda k: runs[k][0])
    v, steps, dt = runs[best]
    return {"value": round(v, 1), "unit": "samples/s", "cores": best, "kind": "port",
            "sample": "%d steps of the same bs-%d joint_elbo train step (oracle/"
                      "mopoe_oracle.py: PyTorch-CPU float32 restatement of the reference "
                      "step, autograd backward, Adam) in %.1f s with %d thread(s); host has "
                      "%d logical CPUs, %d usable by this process" % (
                          steps, c["batch"], dt, best, os.cpu_count() or 0, allc),
            "ms_per_step": round(1e3 * dt / steps, 3),
            "by_threads": {str(k): {"samples_per_s": round(v[0], 1),
                                    "ms_per_step": round(1e3 * v[2] / v[1], 3), "steps": v[1]}
                           for k, v in runs.items()}}


def eager_rocm_baseline(device, seconds=6.0):
    """configs[1]'s "HIP kernels vs PyTorch-ROCm eager": the SAME restatement of the
    reference step as cpu_baseline, with its tensors on the GPU -- every torch op one or
    more device launches (a baseline leg only: the oracle is not the product)."""
    c = CONFIGS["C1"]
    mo, cfg = oracle_setup(c)
    g = torch.Generator().manual_seed(1234)
    pool = [{n: torch.randn(c["batch"], d, generator=g).to(device)
             for n, d in zip(c["names"], c["dims"])} for _ in range(8)]
    shapes = [(c["batch"], LATENT)] + [(c["batch"], s) for s in c["style"]]
    tape = [torch.randn(s, generator=g).to(device) for s in shapes]
    with torch.device(device):
        params = type(mo.init_params(cfg, 0))((k, v.to(device)) for k, v in
                                              mo.init_params(cfg, 0).items())
        state = mo.adam_init(params)
        for i in range(5):
            mo.train_step(params, cfg, pool[i % 8], mo.Noise(tape=tape), state)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        steps = 0
        while time.perf_counter() - t0 < seconds:
            mo.train_step(params, cfg, pool[steps % 8], mo.Noise(tape=tape), state)
            steps += 1
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return {"value": round(c["batch"] * steps / dt, 1), "unit": "samples/s",
            "ms_per_step": round(1e3 * dt / steps, 3), "steps": steps,
            "what": "oracle/mopoe_oracle.py (PyTorch float32 restatement of the reference "
                    "step: autograd backward, hand-written torch-semantics Adam) with all "
                    "tensors on the GPU, PyTorch-ROCm eager, %s" % torch.__version__}


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
    args = parse_args()

    # stdout carries ONE JSON line: whatever libraries print there while the run lasts (RCCL
    # greets with a version banner on stdout when it sets up a communicator) goes to stderr
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py --gpus %d inside a rank environment of WORLD_SIZE=%d" % (args.gpus, world))
    dist = None
    if world > 1 or args.force_dist:
        # (no rank environment and one rank: the launcher above has started this process with
        #  one; under torch.distributed.run the driver's environment is used as it is)
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    c = CONFIGS[args.config]
    if args.global_batch:
        if args.global_batch % world or args.global_batch < world:
            sys.exit("--global-batch %d does not split over %d ranks" % (args.global_batch, world))
        c = dict(c, batch=args.global_batch // world)
    BATCH = c["batch"]      # rows per rank and step
    spec = make_spec(c)
    eng = mm.MoPoEEngine(spec, device, seed=1234)
    eng.reset_parameters(torch.Generator().manual_seed(0))  # same on all ranks
    pool = make_pool(c, device, count=min(POOL, max(2, (1 << 28) // (4 * BATCH * sum(c["dims"])))))
    log_ring = [torch.empty(mm._lib.NUM_STATS, dtype=torch.float32).pin_memory()
                for _ in range(8)]
    fused = dist is None

    # ---- the gradient exchange of the N-rank step
    comm, rccl, exchange, why, in_backward = None, None, "none", "", False
    if dist is not None:
        exchange = os.environ.get("MOPOE_EXCHANGE", "rccl")
        if exchange not in ("auto", "xgmi", "rccl", "c10d"):
            sys.exit("MOPOE_EXCHANGE must be rccl (default), c10d, xgmi or auto")
        if exchange in ("auto", "xgmi"):
            comm, why = open_xgmi(spec.num_floats, device, rank, world, dist)
            if comm is None and exchange == "xgmi":
                sys.exit("MOPOE_EXCHANGE=xgmi but: " + why)
            exchange = "xgmi" if comm is not None else "rccl"
        if comm is not None and os.environ.get("MOPOE_EXCHANGE_AFTER") is None:
            in_backward = check_in_backward(comm, spec, pool[rank % len(pool)], device, world, dist)
            if not in_backward:
                why = "exchange inside the weight-gradient launch failed its start-up check"
        # replicas start identical (parameters, moments, step counts)
        for t in (eng.params, eng.exp_avg, eng.exp_avg_sq, eng.counters):
            dist.broadcast(t, 0)
        eng.refresh_wfrag()      # (c10d writes do not bump tensor._version)
        if exchange == "rccl":
            # the library's own communicator; a set-up that fails (on every rank together) or
            # an all-reduce that does not give c10d's sum leaves the spelled-out form
            try:
                rccl = mm.comm.RcclComm()
                probe = torch.arange(4096, device=device, dtype=torch.float32) * (rank + 1)
                want = probe.clone()
                rccl.allreduce_(probe)
                dist.all_reduce(want)
                ok = torch.tensor([float(torch.equal(probe, want))], device=device)
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if not ok.item():
                    rccl.close()
                    rccl, why = None, "the library's RCCL all-reduce failed its start-up check"
            except mm._lib.MopoeError as e:
                rccl, why = None, "RCCL communicator: %s" % e
            if rccl is None:
                exchange = "c10d"

    def step(i):
        # the step's scalar log lands in a ring of pinned host buffers, written
        # by the kernel itself (no copy on the stream)
        if comm is not None and in_backward:
            return eng.train_step(pool[(i * world + rank) % len(pool)], apply_adam=True, comm=comm,
                                  stats_host=None if args.no_log_copy else log_ring[i % 8])[1]
        if rccl is not None:        # ONE host call: backward, ncclAllReduce, Adam
            return eng.train_step(pool[(i * world + rank) % len(pool)], apply_adam=True, rccl=rccl,
                                  stats_host=None if args.no_log_copy else log_ring[i % 8])[1]
        plan, ws = eng.train_step(pool[(i * world + rank) % len(pool)], apply_adam=fused,
                                  stats_host=None if args.no_log_copy else log_ring[i % 8])
        if comm is not None:
            comm.allreduce_adam(eng)                # one launch: push, sum, Adam
        elif not fused:
            dist.all_reduce(eng.grads)              # RCCL, one flat buffer
            eng.adam_step(world=world)              # mean + the ranks' modality check
        return ws

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    pos = {"n": 0}       # steps enqueued so far (the batch and the log slot of a step follow from it)

    def timed(warmup, steps):
        ws = None
        for _ in range(warmup):
            step(pos["n"])
            pos["n"] += 1
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            ws = step(pos["n"])
            pos["n"] += 1
        barrier()
        return time.perf_counter() - t0, ws

    # Timed first thing (a chip that has idled: clocks still ramping, caches and TLBs cold --
    # reported as `cold_start`), then again after SETTLE untimed steps: the W warm-up steps and
    # EXACTLY K timed steps as asked, on a chip at its working clocks -- `value`, what a training
    # run sees after its first few milliseconds.  A region of 20 steps is 0.6 ms between two
    # synchronisations: on ONE box four runs of the driver's invocation read 30.7, 31.3, 35.5 and
    # 42.7 us per step (round 4), so for K <= 200 the region is timed REPS = 5 times back to back
    # (each exactly K steps between barrier + synchronize pairs) and `value` is the MEDIAN; every
    # sample is in the line (`value_samples_ms_per_step`).
    SETTLE = args.settle
    REPS = 5 if args.steps <= 200 else 1
    cold = {}

    def measure():
        dt0, _ = timed(args.warmup, args.steps)
        cold["dt"] = dt0
        # (the settling steps are timed as one region too: `long_run` -- what a K of thousands
        #  reads; a 20-step region of 0.7 ms carries its two synchronisations, ~80 us)
        cold["settle_dt"], _ = timed(0, SETTLE)
        samples, ws = [], None
        for r in range(REPS):
            dt_r, ws = timed(args.warmup if r == 0 else 0, args.steps)
            samples.append(dt_r)
        cold["samples"] = samples
        return sorted(samples)[len(samples) // 2], ws

    dt, ws = measure()
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
            comm, exchange = None, "c10d"
            eng.recover()
            eng.reset_parameters(torch.Generator().manual_seed(0))
            eng.exp_avg.zero_()
            eng.exp_avg_sq.zero_()
            dt, ws = measure()
    replicas_identical = None
    if dist is not None:
        # (every form: the replicas must hold the same parameters after the timed region)
        mine = eng.params.double().sum()
        lo, hi = mine.clone(), mine.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        replicas_identical = bool(lo.item() == hi.item())
    if dist is not None:
        t = torch.tensor([cold["dt"], cold["settle_dt"]] + cold["samples"], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)     # (every region: the slowest rank's time)
        v = [float(x) for x in t.tolist()]
        cold["dt"], cold["settle_dt"], cold["samples"] = v[0], v[1], v[2:]
        dt = sorted(cold["samples"])[len(cold["samples"]) // 2]
    loss = float(ws.stats[0].item())
    eng.check_valid(sync=True)     # (raises if any step of the timed region was invalid)
    if dist is not None:           # replicas must still be identical
        mine = eng.params.double().sum()
        lo, hi = mine.clone(), mine.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        if lo.item() != hi.item():
            sys.exit("data-parallel replicas drifted apart")
    if not args.no_log_copy:   # the host ring must have received the same scalar
        host = float(log_ring[(pos["n"] - 1) % 8][0])
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
        "higher_is_better": True, "scaling": "strong" if args.global_batch else "weak",
        "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": c["label"],
                   "global_batch": BATCH * world, "batch_per_rank": BATCH,
                   "parallelism": "dp%d" % world if dist is not None else "single",
                   "launched_by": ("bench.py's own launcher (child processes)"
                                   if os.environ.get("MOPOE_BENCH_SPAWNED") else
                                   "torch.distributed.run" if "WORLD_SIZE" in os.environ else
                                   "single process"),
                   "exchange_form": exchange,
                   "rccl": ({"world": rccl.world, "rank": rccl.rank} if rccl is not None else None),
                   "exchange": {"none": "none (single GPU: Adam fused into the "
                                        "weight-gradient launch)",
                                "xgmi": ("xGMI peer windows inside the weight-gradient launch: "
                                         "every gradient block is pushed to every peer, summed "
                                         "in rank order, Adam applied" if dist is not None and
                                         comm is not None and in_backward else
                                         "one launch per rank over xGMI peer windows: push "
                                         "to every peer, rank-ordered sum, Adam"),
                                "rccl": "ONE host call per step (mopoe_rccl_train_step): "
                                        "backward, ncclAllReduce of the flat gradient buffer "
                                        "over the library's own RCCL communicator, Adam "
                                        "kernel (mean, the ranks' modality / validity check)",
                                "c10d": "three host calls per step: train_step, "
                                        "torch.distributed all_reduce (RCCL), adam_step"
                                }[exchange] + (" [%s]" % why if why else ""),
                   "replicas_identical_after_timed_region": replicas_identical,
                   "host_log_every_step": not args.no_log_copy,
                   "host_wait": "polling (HSA_ENABLE_INTERRUPT=0)"
                                if os.environ.get("HSA_ENABLE_INTERRUPT") == "0" else "interrupts",
                   "settle_steps": SETTLE,
                   # what ran before the region(s) `value` is read from
                   "warmup_effective": 2 * args.warmup + args.steps + SETTLE,
                   "timing": ("median of %d regions of exactly %d steps, each between barrier + "
                              "synchronize pairs, back to back" % (REPS, args.steps)) if REPS > 1 else
                             "one region of exactly %d steps" % args.steps,
                   "final_loss": round(loss, 3)},
        "cold_start": {"ms_per_step": round(1e3 * cold["dt"] / args.steps, 5),
                       "value": round(BATCH * world * args.steps / cold["dt"], 1),
                       "unit": "samples/s",
                       "what": "the same W warm-up + K timed steps as the first GPU work of "
                               "the process (idle clocks, cold caches); `value` is timed after "
                               "%d more untimed steps" % SETTLE},
    }
    if SETTLE > 0:
        out["long_run"] = {"steps": SETTLE, "ms_per_step": round(1e3 * cold["settle_dt"] / SETTLE, 5),
                           "value": round(BATCH * world * SETTLE / cold["settle_dt"], 1),
                           "unit": "samples/s",
                           "what": "the settling steps timed as one region (same loop, same "
                                   "barriers): the figure a K of thousands reads"}

    out["value_samples_ms_per_step"] = [round(1e3 * x / args.steps, 5) for x in cold["samples"]]
    nxt = pos["n"]
    if rank == 0 and not args.no_roofline:
        # instrumented re-run of the same K steps: HIP events around every
        # launch, on the launch stream
        prof = profile_steps(step, nxt, args.steps)
        # (the event overhead is what the event-timed kernels of a step sum to beyond the step
        #  itself: the step time of the longest region timed is the one to hold them against --
        #  a 20-step region carries 4 us per step of its two synchronisations)
        steady = dt / args.steps
        if SETTLE >= 10 * args.steps:
            steady = min(steady, cold["settle_dt"] / SETTLE)
        out["roofline"] = roofline_of(spec, BATCH, prof,
                                      steady if (dist is None and fused) else None,
                                      fused or in_backward, world, traffic=args.config)
    elif dist is not None and not args.no_roofline:
        for i in range(args.steps):   # keep the collectives matched
            step(nxt + i)
        torch.cuda.synchronize()
    nxt += args.steps

    single = rank == 0 and world == 1 and dist is None
    if single and not args.quick:
        out["other_configs"] = {k: other_config(k, device) for k in ("C3", "C5")}
        out["loop"] = {"what": "run_epochs.train over a ResidentCohort (sampler + index "
                               "batches + fused steps + per-step validity look), not the "
                               "headline: the drop-in loop a user of the reference gets",
                       "cohorts": [loop_figure(device, 3000), loop_figure(device, 16384)]}
        for c_ in out["loop"]["cohorts"]:
            c_["ratio_to_bare_engine_loop"] = round(c_["us_per_step"] / (1e3 * out["ms_per_step"]), 3)
        out["regime_n65536"] = regime_point(device)
        out["regime_n65536_bf16_operands"] = regime_point(device, operands="bf16")
        out["eager_rocm_baseline"] = eager_rocm_baseline(device)
    if single and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    if comm is not None:
        comm.close()
    if rccl is not None:
        rccl.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
