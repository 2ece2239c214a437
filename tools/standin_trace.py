"""A few training steps at the benchmark geometry with a stand-in collective per gradient bucket (tools/stage_cost.py leg d),
for `rocprofv3 --kernel-trace`: which kernels wait for the compute units the stand-in holds, and for how long.
usage: python tools/standin_trace.py <workgroups> <lds_bytes> <vgprs 0|128> <usec> [cu_budget]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "h-denseformer_amd")):
    sys.path.insert(0, p)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29593")
import torch
import torch.distributed as dist

wgs, lds, vg, usec = [int(v) for v in sys.argv[1:5]]
budget = int(sys.argv[5]) if len(sys.argv) > 5 else 256
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=0, world_size=1)
from hdf_rt._lib import check, lib
from hdf_rt.optim import FlatAdam
from hdf_rt.parallel import GradSync
from loss.combine_loss import CEPlusDice, DeepSuperloss
from models.HDenseFormer import HDenseFormer


class OccupySync(GradSync):
    def _reduce(self, chunk):
        us = max(10, int(usec * chunk.numel() * 4 / 26.0e6))     # `usec` per 26 MB (the largest bucket of rounds 4-5)
        check(lib().hdf_op_occupy(wgs, lds, vg, us, torch.cuda.current_stream().cuda_stream), "occupy")


dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = HDenseFormer(4, 4, 32, image_size=(128,) * 3, transformer_depth=24).to(dev)
net.train()
net.compute_dtype = "bf16"
crit = DeepSuperloss(criterion=CEPlusDice(weight=None, ignore_index=0))
opt = FlatAdam(net, lr=1e-3, weight_decay=1e-4)
g = torch.Generator(device="cpu").manual_seed(1234)
x = torch.rand(2, 4, 128, 128, 128, generator=g).to(dev)
lab = torch.randint(0, 4, (2, 128, 128, 128), generator=g)
t = torch.nn.functional.one_hot(lab, 4).permute(0, 4, 1, 2, 3).float().contiguous().to(dev)
check(lib().hdf_set_cu_budget(budget), "budget")
sync = OccupySync(net) if wgs > 0 else None
net.grad_hook = sync
evs = []
for k in range(8):
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record()
    evs.append(e0)
    opt.zero_grad()
    crit(net(x), t).backward()
    if sync is not None:
        sync.wait()
    opt.step()
e0 = torch.cuda.Event(enable_timing=True)
e0.record()
evs.append(e0)
torch.cuda.synchronize()
print("ms per step:", [round(evs[k].elapsed_time(evs[k + 1]), 3) for k in range(8)])
dist.destroy_process_group()
