"""Diagnostic: what a training loop that synchronises once per step (loss.item(), like trainer.py:382-400) pays on top of the
device time of a step: wall clock per step with a synchronize after every step against the HIP-event time of the same steps."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "h-denseformer_amd")):
    sys.path.insert(0, p)
import torch
from hdf_rt.optim import FlatAdam
from loss.combine_loss import CEPlusDice, DeepSuperloss
from models.HDenseFormer import HDenseFormer

dev = torch.device("cuda", 0)
net = HDenseFormer(4, 4, 32, image_size=(128, 128, 128), transformer_depth=24).to(dev)
net.train()
net.compute_dtype = "bf16"
crit = DeepSuperloss(criterion=CEPlusDice(weight=None, ignore_index=0))
opt = FlatAdam(net, lr=1e-3, weight_decay=1e-4)
g = torch.Generator().manual_seed(1)
x = torch.rand(2, 4, 128, 128, 128, generator=g).to(dev)
lab = torch.randint(0, 4, (2, 128, 128, 128), generator=g)
target = torch.nn.functional.one_hot(lab, 4).permute(0, 4, 1, 2, 3).float().contiguous().to(dev)


def step():
    opt.zero_grad()
    loss = crit(net(x), target)
    loss.backward()
    opt.step()
    return loss


for _ in range(5):
    step()
torch.cuda.synchronize()
walls, devs, hosts = [], [], []
for _ in range(15):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    loss = step()
    e1.record()
    t1 = time.perf_counter()
    v = loss.item()                     # the trainer's per-step synchronisation
    t2 = time.perf_counter()
    walls.append((t2 - t0) * 1e3)
    hosts.append((t1 - t0) * 1e3)
    devs.append(e0.elapsed_time(e1))
med = lambda a: sorted(a)[len(a) // 2]
print(f"sync-per-step loop: wall {med(walls):.2f} ms/step, host enqueue {med(hosts):.2f} ms, HIP events {med(devs):.2f} ms")
