#!/usr/bin/env python3
"""Same-box A/B of builds and knobs over bench.py's configurations (boxes differ by a few
percent, so variants are only ever compared inside ONE gpurun call, interleaved, ROUNDS times):

    python3 tools/ab_configs.py [--configs C1,C3,C5] [--rounds 3] label[:ENV=VAL[,ENV=VAL...]] ...

e.g.  base  nopipe:MOPOE_LIB=libmopoe_hip_vNoPipe.so  noxcd:MOPOE_WGRAD_XCD=0
Every (round, variant) is a child process (the library and its knobs are read once per
process): us per training step (1,500 steps after 300) and per kernel (HIP events, 500 steps)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(configs, steps):
    sys.path.insert(0, ROOT)
    import torch
    import bench
    for key in configs:
        c = bench.CONFIGS[key]
        dt, eng, step, sp = bench.time_single_gpu(c, torch.device("cuda"), steps, 300)
        prof = bench.profile_steps(step, steps + 300, 500)
        print("%s %8.2f us/step | %s" % (key, 1e6 * dt / steps, "  ".join(
            "%s %.2f" % (k, v[1] / v[0] * 1e3) for k, v in prof.items() if v[0])), flush=True)


def main():
    argv = sys.argv[1:]
    configs, rounds, steps = ["C1", "C3", "C5"], 3, 1500
    variants = []
    while argv:
        a = argv.pop(0)
        if a == "--configs":
            configs = argv.pop(0).split(",")
        elif a == "--rounds":
            rounds = int(argv.pop(0))
        elif a == "--steps":
            steps = int(argv.pop(0))
        elif a == "--child":
            return child(argv.pop(0).split(","), int(argv.pop(0)))
        else:
            label, _, envs = a.partition(":")
            variants.append((label, dict(e.split("=", 1) for e in envs.split(",") if e)))
    for r in range(rounds):
        for label, env in variants:
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", ",".join(configs), str(steps)],
                               env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            if p.returncode:
                print("round %d %-10s FAILED rc %d: %s" % (r, label, p.returncode, p.stderr[-400:]), flush=True)
                continue
            for line in p.stdout.splitlines():
                if "us/step" in line:
                    print("round %d %-10s %s" % (r, label, line), flush=True)


if __name__ == "__main__":
    main()
