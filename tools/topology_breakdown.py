import os, sys
sys.path.insert(0, "/root/repo")
import mopoe_amd as mm, torch, bench
c = bench.CONFIGS["C1"]
for label, kw, env in (("default quad", {}, {}), ("default 16-row lean", {}, {"MOPOE_QUAD": "0"}),
                       ("default 16-row generic", {}, {"MOPOE_QUAD": "0", "MOPOE_NO_LEAN": "1"}),
                       ("sample scale", dict(sample_scale=True), {}), ("enc 0", dict(enc_layers=0), {})):
    for k in ("MOPOE_QUAD", "MOPOE_NO_LEAN"):
        os.environ.pop(k, None)
    os.environ.update(env)
    mm._lib.reload_knobs()
    spec = mm.ModelSpec(c["names"], c["dims"], c["style"], class_dim=20, method="joint_elbo", **kw)
    eng = mm.MoPoEEngine(spec, "cuda", seed=1)
    eng.reset_parameters(torch.Generator().manual_seed(0))
    pool = bench.make_pool(c, torch.device("cuda"), count=8)
    for i in range(300): eng.train_step(pool[i % 8])
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for i in range(1500): eng.train_step(pool[i % 8])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 1500
    mm._lib.profile_enable(True)
    for i in range(300): eng.train_step(pool[i % 8])
    torch.cuda.synchronize()
    prof = mm._lib.profile_read(); mm._lib.profile_enable(False)
    print("%-26s %6.1f us/step | %s" % (label, 1e6 * dt, "  ".join("%s %.2f" % (k, v[1] / v[0] * 1e3) for k, v in prof.items() if v[0])), flush=True)
