#!/usr/bin/env python3
"""Diagnostic: does replaying the step's three launches out of a hipGraph shorten the step?
Captures POOL consecutive train steps (one per resident batch) into one graph and times
replays against the same steps launched eagerly."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mopoe_amd as mm

N, POOL = 256, 64
spec = mm.ModelSpec(["clinical", "rois"], [7, 444], [3, 20], method="joint_elbo")
eng = mm.MoPoEEngine(spec, "cuda", seed=1)
g = torch.Generator().manual_seed(0)
pool = [{"clinical": torch.randn(N, 7, generator=g).cuda(),
         "rois": torch.randn(N, 444, generator=g).cuda()} for _ in range(POOL)]
ring = [torch.empty(mm._lib.NUM_STATS, dtype=torch.float32).pin_memory() for _ in range(8)]
for i in range(200):
    eng.train_step(pool[i % POOL], stats_host=ring[i % 8])
torch.cuda.synchronize()
t0 = time.perf_counter()
for r in range(20):
    for i in range(POOL):
        eng.train_step(pool[i], stats_host=ring[i % 8])
torch.cuda.synchronize()
eager = (time.perf_counter() - t0) / (20 * POOL)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        for i in range(POOL):
            eng.train_step(pool[i], stats_host=ring[i % 8])
torch.cuda.synchronize()
for r in range(3):
    graph.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for r in range(20):
    graph.replay()
torch.cuda.synchronize()
rep = (time.perf_counter() - t0) / (20 * POOL)
print("eager  %.2f us/step" % (eager * 1e6))
print("graph  %.2f us/step   loss %.3f" % (rep * 1e6, float(ring[(POOL - 1) % 8][0])))
