import sys, torch
sys.path[:0] = [".", "tests", "oracle"]
import mopoe_oracle as mo, mopoe_amd as mm
from hip_util import make_engine
C5 = dict(names=["clinical", "rois", "snps", "tracts"], input_dim=[7, 444, 128, 64], style_dim=[3, 3, 3, 3])
cfg = mo.Config(**C5)
for n in (512, 2048, 4096):
    spec, eng = make_engine(cfg)
    x = mo.make_inputs(cfg.names, cfg.input_dim, n, seed=3)
    noise = mo.Noise(generator=mo.noise_rng(4))
    mo.forward(mo.init_params(cfg, 0), cfg, x, noise)
    plan, ws = eng.train_step(x, eps=noise.tape, apply_adam=False)
    torch.cuda.synchronize()
    for m, name in enumerate(cfg.names):
        wh = eng.params[spec.c_model.off_wh[m]:spec.c_model.off_wh[m] + spec.heads_dim(m) * 256].view(-1, 256)
        ref = (ws.g_heads[m] @ wh) * (ws.hidden[m] > 0)
        err = (ws.g_pre[m] - ref).abs()
        bad = (err > 1e-6 * ref.abs().max()).nonzero()
        print(n, name, "g_pre max err %.3e of %.3e; bad %d" % (err.max().item(), ref.abs().max().item(), len(bad)),
              "rows", sorted(set(bad[:, 0].tolist()))[:8], "cols", sorted(set(bad[:, 1].tolist()))[:8])
        xt = x[name].cuda()
        gw = ws.g_pre[m].t() @ xt
        got = eng.grad_views["encoders.%s.shared_encoder.0.weight" % name]
        print("    W1 grad vs g_pre^T x: %.3e" % (got - gw).abs().max().item())
