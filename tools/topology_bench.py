#!/usr/bin/env python3
"""Diagnostic: the training step of the topologies train_exp exposes beyond its defaults
(hidden encoder / decoder layers, dropout, learn_output_sample_scale; DESIGN.md section
3b) next to the default topology, configs[1]'s modalities and batch: us per step with
the device-drawn noise and dropout masks."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib.util  # noqa: E402

_spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(
    os.path.dirname(os.path.abspath(__file__))), "bench.py"))
bench = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(bench)
import torch  # noqa: E402

mm = bench.mm

TOPOLOGIES = [
    ("default (1 hidden encoder layer)", {}),
    ("enc 2, dec 1", dict(enc_layers=2, dec_layers=1)),
    ("enc 2, dec 1, dropout 0.2", dict(enc_layers=2, dec_layers=1, dropout=0.2)),
    ("enc 3, dec 2, dropout 0.2", dict(enc_layers=3, dec_layers=2, dropout=0.2)),
    ("enc 1, dec 0, sample scale", dict(sample_scale=True)),
    ("enc 0", dict(enc_layers=0)),
    ("enc 2, dec 1, dropout 0.2, method poe", dict(enc_layers=2, dec_layers=1, dropout=0.2,
                                                    method="poe")),
]


def main():
    device = torch.device("cuda")
    steps, warmup = 400, 100
    only = os.environ.get("TOPOLOGY_ONLY")      # (profiling: one topology, by its label)
    for key in os.environ.get("TOPOLOGY_CONFIGS", "C1,C5").split(","):
        c = bench.CONFIGS[key]
        print(c["label"])
        base = None
        for label, topo in TOPOLOGIES:
            if only and label != only:
                continue
            kw = dict(topo)
            method = kw.pop("method", c["method"])
            spec = mm.ModelSpec(c["names"], c["dims"], c["style"], class_dim=bench.LATENT,
                                method=method, **kw)
            eng = mm.MoPoEEngine(spec, device, seed=1234)
            eng.reset_parameters(torch.Generator().manual_seed(0))
            pool = bench.make_pool(c, device, count=16)
            for i in range(warmup):
                eng.train_step(pool[i % len(pool)])
            torch.cuda.synchronize()
            # (the median of three regions: a chain of 12-20 launches has a host side of 50-100 us
            #  per step, and one hiccup of the host inside a single 60 ms region reads as +10-50 %)
            regions = []
            for _ in range(3):
                t0 = time.perf_counter()
                for i in range(steps):
                    eng.train_step(pool[i % len(pool)])
                torch.cuda.synchronize()
                regions.append(1e6 * (time.perf_counter() - t0) / steps)
            us = sorted(regions)[1]
            eng.check_valid(sync=True)
            if base is None:
                base = us
            print("  %-40s %8.1f us/step  %6.2f M samples/s  %5.1f x default" % (
                label, us, c["batch"] / us, us / base))
            del eng, pool


if __name__ == "__main__":
    main()
