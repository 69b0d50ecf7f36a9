import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import mopoe_amd as mm
n=256
spec = mm.ModelSpec(["clinical", "rois"], [7, 444], [3, 20], method="joint_elbo")
eng = mm.MoPoEEngine(spec, "cuda", seed=1)
g = torch.Generator().manual_seed(0)
pool = [{"clinical": torch.randn(n, 7, generator=g).cuda(), "rois": torch.randn(n, 444, generator=g).cuda()} for _ in range(8)]
for i in range(300): eng.train_step(pool[i % 8])
torch.cuda.synchronize()
lo, hi = int(sys.argv[1]), int(sys.argv[2])
rows=[]
for it in range(60):
    plan, ws = eng.train_step(pool[it % 8]); torch.cuda.synchronize()
    s = ws._stats_all.cpu().view(torch.int32).numpy().astype(np.int64) & 0xFFFFFFFF
    rows.append(s[64+lo:64+hi+1])
r=np.array(rows); rel=((r-r[:,:1])%(1<<32))/100.0
print(np.round(np.median(rel,axis=0),2))
