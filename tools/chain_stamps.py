#!/usr/bin/env python3
"""Diagnostic: stage stamps of the general chain's element-wise kernels (the -DMOPOE_STAMPS build:
make -C 2022_cambroise_interpret_multivae_amd/csrc stamps): block 0, thread 0, 100 MHz clock."""
import os
import sys

os.environ.setdefault("MOPOE_LIB", "libmopoe_hip_stamps.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

mm = bench.mm
c = bench.CONFIGS["C1"]
spec = mm.ModelSpec(c["names"], c["dims"], c["style"], class_dim=bench.LATENT, method=c["method"],
                    enc_layers=2, dec_layers=1, dropout=0.2)
eng = mm.MoPoEEngine(spec, "cuda", seed=1)
eng.reset_parameters(torch.Generator().manual_seed(0))
pool = bench.make_pool(c, torch.device("cuda"), count=8)
for i in range(200):
    eng.train_step(pool[i % 8])
torch.cuda.synchronize()
LO, HI = 40, 56
acc = torch.zeros(HI - LO - 1, dtype=torch.float64)
for it in range(50):
    eng.train_step(pool[it % 8])
    torch.cuda.synchronize()
    raw = eng.counters[64 + LO:64 + HI].cpu().double()
    acc += ((raw[1:] - raw[:-1]) % 4294967296.0)
for i, v in enumerate(acc / 50):
    print("stamp %d -> %d: %7.2f us" % (LO + i, LO + i + 1, v / 100.0))
