#!/usr/bin/env python3
"""Diagnostic: where k_latent spends its time.  Uses the -DMOPOE_STAMPS build
(make -C 2022_cambroise_interpret_multivae_amd/csrc stamps), which must report
ScratchSize 0 for the numbers to mean anything:
    python tools/stage_stamps.py [method] [N]
Prints per-stage durations of block 0 (100 MHz s_memrealtime) and the shader
clock it ran at (delta s_memtime / delta s_memrealtime x 100 MHz)."""
import os
import sys

os.environ.setdefault("MOPOE_LIB", "libmopoe_hip_stamps.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import mopoe_amd as mm  # noqa: E402

method = sys.argv[1] if len(sys.argv) > 1 else "joint_elbo"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
if method in ("C1", "C3", "C5"):        # bench.py's configurations
    import bench
    cfg = bench.CONFIGS[method]
    spec = bench.make_spec(cfg)
    n = int(sys.argv[2]) if len(sys.argv) > 2 else cfg["batch"]
    names_, dims_ = cfg["names"], cfg["dims"]
else:
    spec = mm.ModelSpec(["clinical", "rois"], [7, 444], [3, 20], method=method)
    names_, dims_ = ["clinical", "rois"], [7, 444]
eng = mm.MoPoEEngine(spec, "cuda", seed=1)
g = torch.Generator().manual_seed(0)
pool = [{k: torch.randn(n, d, generator=g).cuda() for k, d in zip(names_, dims_)} for _ in range(8)]
names = ["S0 h->LDS", "S1 heads", "S2a combine + x->LDS", "S2b fusion fwd",
         "S3 decoder+NLL (+KL sums)", "S4 g_z partials", "S4b (poe only)", "S5 fusion bwd",
         "S6 g_pre", "tail"]
used = [0, 1, 2, 3, 4, 6, 7, 8, 9, 10, 11]   # stamp 5 is no longer taken
for i in range(300):
    plan, ws = eng.train_step(pool[i % 8])
torch.cuda.synchronize()
acc = None
for it in range(40):
    plan, ws = eng.train_step(pool[it % 8])
    torch.cuda.synchronize()
    raw = ws._stats_all[128:128 + 30].cpu().view(torch.int32).view(15, 2).double()
    st = raw[used]
    d = (st[1:] - st[:-1]) % 4294967296.0
    acc = d if acc is None else acc + d
    # inside S3, wave 0: stamps 12 (MFMAs issued), 13 (rows done), 14 (at the barrier)
    x = torch.stack([raw[12] - raw[4], raw[13] - raw[12], raw[14] - raw[13],
                     raw[6] - raw[14]]) % 4294967296.0
    inner = x if it == 0 else inner + x
    x4 = torch.stack([raw[12] - raw[6], raw[13] - raw[12], raw[14] - raw[13],
                      raw[7] - raw[14]]) % 4294967296.0
    inner4 = x4 if it == 0 else inner4 + x4
acc /= 40
inner /= 40
inner4 /= 40
print("k_latent stage times, %s, N=%d (block 0)" % (method, n))
for nme, row in zip(names, acc):
    rt, mt = row[0].item(), row[1].item()
    print("%-22s %6.2f us   clock %5.0f MHz" % (nme, rt / 100.0, (mt / rt * 100.0) if rt else 0))
print("%-22s %6.2f us" % ("total", acc[:, 0].sum().item() / 100.0))
if os.environ.get("MOPOE_QUAD") == "1":
    print("(four-row form) inside S4, wave 0: scalar set-up %.2f | reads + MFMAs %.2f | partial stores %.2f | "
          "to the barrier's end %.2f us" % tuple(inner4[:, 0] / 100.0))
print("inside S3 (wave 8): loads+MFMA issue %.2f | four rows of epilogue %.2f | sums %.2f | "
      "waiting at the barrier %.2f us" % tuple(inner[:, 0] / 100.0))

