import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mopoe_amd as mm
n=256
spec = mm.ModelSpec(["clinical", "rois"], [7, 444], [3, 20], method="joint_elbo")
eng = mm.MoPoEEngine(spec, "cuda", seed=1)
g = torch.Generator().manual_seed(0)
pool = [{"clinical": torch.randn(n, 7, generator=g).cuda(), "rois": torch.randn(n, 444, generator=g).cuda()} for _ in range(8)]
for i in range(300): eng.train_step(pool[i % 8])
torch.cuda.synchronize()
order = [int(x) for x in sys.argv[1].split(",")]
acc=None
for it in range(40):
    plan, ws = eng.train_step(pool[it % 8]); torch.cuda.synchronize()
    st = ws._stats_all[64:64+32].cpu().view(torch.int32).view(16,2).double()[order]
    d = (st[1:]-st[:-1]) % 4294967296.0
    acc = d if acc is None else acc+d
acc/=40
for a,b,row in zip(order[:-1],order[1:],acc): print("%2d -> %2d  %6.2f us  (%5.0f shader cycles)"%(a,b,row[0].item()/100,row[1].item()))
