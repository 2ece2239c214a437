"""A few train steps of HDenseFormer_2D_32 at the reference's own 2-D workload (PI-CAI: 2x384^2, batch 24, td 16, bf16) for
`rocprofv3 --kernel-trace --stats`: where the native depth-1 path (round 6) spends its step."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "h-denseformer_amd")):
    sys.path.insert(0, p)
import torch

from hdf_rt.optim import FlatAdam
from loss.combine_loss import CEPlusDice, DeepSuperloss
from models.HDenseFormer_2D import HDenseFormer_2D_32

dev = "cuda:0"
g = torch.Generator().manual_seed(3)
net = HDenseFormer_2D_32(2, 2, (384, 384), 16).to(dev)
net.train()
net.compute_dtype = "bf16"
if len(sys.argv) > 1 and sys.argv[1] == "embedded":
    net._embedded_2d = True
x = torch.rand(24, 2, 384, 384, generator=g).to(dev)
t = torch.nn.functional.one_hot(torch.randint(0, 2, (24, 384, 384), generator=g), 2).permute(0, 3, 1, 2).float().contiguous().to(dev)
crit = DeepSuperloss(criterion=CEPlusDice(weight=None, ignore_index=0))
opt = FlatAdam(net, lr=1e-3, weight_decay=1e-4)
evs = []
for k in range(6):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    evs.append(e)
    opt.zero_grad()
    crit(net(x), t).backward()
    opt.step()
e = torch.cuda.Event(enable_timing=True)
e.record()
evs.append(e)
torch.cuda.synchronize()
print("ms per step:", [round(evs[k].elapsed_time(evs[k + 1]), 3) for k in range(6)])
