import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle"); sys.path.insert(0, "/root/repo/tests")
import mopoe_oracle as mo
from hip_util import make_engine
cfg = mo.Config(["clinical", "rois"], [7, 444], [3, 20])
spec, eng = make_engine(cfg)
x = mo.make_inputs(cfg.names, cfg.input_dim, 32, seed=11)
noise = mo.Noise(generator=mo.noise_rng(12))
out, grads = mo.loss_and_grads(mo.init_params(cfg, 0), cfg, x, noise)
plan, ws = eng.train_step(x, eps=noise.tape, apply_adam=False)
torch.cuda.synchronize()
for m, name in enumerate(cfg.names):
    loc = ws.loc[m][:32].cpu(); gx = ws.g_xhat[m][:32].cpu()
    want = -(x[name] - loc) * 20.0855369 / 32
    err = (gx - want).abs()
    print(name, "g_xhat max err", err.max().item(), "bad cols:", (err.max(0).values > 1e-4).nonzero().flatten().tolist()[:40], "bad rows:", (err.max(1).values > 1e-4).nonzero().flatten().tolist())
    bad = (err > 1e-4)
    if bad.any():
        i = bad.nonzero()[0]
        r, c = int(i[0]), int(i[1])
        print("  first bad (row %d col %d): got %g want %g  x=%g loc=%g -> implied x = %g" % (r, c, gx[r,c], want[r,c], x[name][r,c], loc[r,c], loc[r,c] - gx[r,c]*32/20.0855369))
        print("  x row:", x[name][r,:8].tolist())
        print("  implied x row:", (loc[r] - gx[r]*32/20.0855369)[:8].tolist())
