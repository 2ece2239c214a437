"""Sustained timing of the InstanceNorm(+ReLU) backward op (reduce + finalize + apply) through the C ABI."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "h-denseformer_amd")):
    sys.path.insert(0, p)
import torch

from hdf_rt._lib import BF16, check, lib, ptr

n, c, s = 2, int(sys.argv[1]) if len(sys.argv) > 1 else 32, int(sys.argv[2]) if len(sys.argv) > 2 else 128
vox = s ** 3
dev = "cuda:0"
st = torch.cuda.current_stream().cuda_stream
da = torch.randn(n, vox, c, device=dev).to(torch.bfloat16)
y = torch.randn(n, vox, c, device=dev).to(torch.bfloat16)
dy = torch.empty_like(y)
sc, sh, mu, rs = (torch.rand(n, c, device=dev) + 0.5 for _ in range(4))
gamma = torch.ones(c, device=dev)
dg, db = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
ws = torch.empty(lib().hdf_op_in_bwd_workspace_floats(n, c, vox), device=dev)


def launch():
    check(lib().hdf_op_in_bwd(BF16, ptr(da), c, ptr(y), c, ptr(sc), ptr(sh), ptr(mu), ptr(rs), ptr(gamma), ptr(dy), c,
                              ptr(dg), ptr(db), n, c, vox, ptr(ws), st), "in_bwd")


for _ in range(5):
    launch()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
reps = 50
for _ in range(reps):
    launch()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
gb = n * vox * c * 2 * 5 / 1e9
print(f"in_bwd C={c} @{s}^3 n={n}: {ms*1e3:.0f} us total (5 passes over the tensor = {gb:.2f} GB -> {gb/ms:.2f} TB/s)")
