import sys
sys.path[:0] = ["tests", ".", "oracle"]   # (a diagnostic beside the tests: the oracle is their checker)
from collections import OrderedDict
import torch
import mopoe_oracle as mo
from hip_util import make_engine
C5 = dict(names=["clinical", "rois", "snps", "tracts"], input_dim=[7, 444, 128, 64], style_dim=[3, 3, 3, 3])
cfg = mo.Config(method="joint_elbo", **C5, sample_scale=True)
spec, eng = make_engine(cfg)
for step in range(5):
    x = mo.make_inputs(cfg.names, cfg.input_dim, 200, seed=40 + step)
    noise = mo.Noise(generator=mo.noise_rng(50 + step))
    params_now = OrderedDict((k, v.cpu().clone()) for k, v in eng.named_params().items())
    st = {"step": OrderedDict((k, step) for k in params_now),
          "exp_avg": OrderedDict((k, v.cpu().clone()) for k, v in spec.param_views(eng.exp_avg).items()),
          "exp_avg_sq": OrderedDict((k, v.cpu().clone()) for k, v in spec.param_views(eng.exp_avg_sq).items())}
    before = OrderedDict((k, v.clone()) for k, v in params_now.items())   # (mo.train_step updates in place)
    out, grads = mo.train_step(params_now, cfg, x, noise, st)
    params_now = before
    plan, ws = eng.train_step(x, eps=noise.tape)
    torch.cuda.synchronize()
    k = "encoders.tracts.shared_encoder.0.weight"
    d = (eng.grad_views[k].cpu() - grads[k]).abs()
    bad = (d.max(1).values > 1e-4).nonzero().flatten().tolist()
    print("step", step, "units off:", bad)
    for u in bad:
        W, b = params_now[k].double(), params_now[k.replace("weight", "bias")].double()
        pre = x["tracts"].double() @ W[u] + b[u]
        i = pre.abs().argmin().item()
        print("   unit", u, "row", i, "pre-activation (float64)", pre[i].item(), "engine act", ws.enc_act[3][0][i, u].item())
        Wh = torch.cat([params_now["encoders.tracts.%s.weight" % h] for h in ("style_mu", "style_logvar", "class_mu", "class_logvar")]).cuda()
        n = 200
        want = (ws.g_heads[3][:n] @ Wh) * (ws.enc_act[3][0][:n] > 0)
        got = ws.g_enc[3][0][:n]
        dd = (want - got).abs()
        print("   g_enc vs recomputed: max err", dd.max().item(), "rows off for this unit", (dd[:, u] > 1e-6).nonzero().flatten().tolist())
        r = dd[:, u].argmax().item()
        print("   row", r, "got", got[r, u].item(), "want", want[r, u].item(), "act", ws.enc_act[3][0][r, u].item(), "pre64", pre[r].item())
