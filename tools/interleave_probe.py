import os, sys
ROOT = "/root/repo"
for p in (ROOT, os.path.join(ROOT, "h-denseformer_amd")):
    sys.path.insert(0, p)
import torch
from hdf_rt._lib import BF16, check, lib, ptr
n, c, s = 2, 32, 128
vox = s ** 3
dev = "cuda:0"
st = torch.cuda.current_stream().cuda_stream
da = torch.randn(n, vox, c, device=dev).to(torch.bfloat16); y = torch.randn(n, vox, c, device=dev).to(torch.bfloat16)
dy = torch.empty_like(y)
sc, sh, mu, rs = (torch.rand(n, c, device=dev) + 0.5 for _ in range(4))
gamma = torch.ones(c, device=dev); dg = torch.zeros(c, device=dev); db = torch.zeros(c, device=dev)
ws = torch.empty(lib().hdf_op_in_bwd_workspace_floats(n, c, vox), device=dev)
cin, cout = 64, 32
x = torch.randn(n, s, s, s, cin, device=dev).to(torch.bfloat16)
w = (torch.randn(27 * 32 * cin, device=dev) * 0.02).to(torch.bfloat16)
out = torch.empty(n, s, s, s, cout, device=dev, dtype=torch.bfloat16)
tiles = lib().hdf_op_conv3d_stat_tiles(BF16, cin, s, s, s)
part = torch.empty(n * tiles * 32 * 2, device=dev)
def conv():
    check(lib().hdf_op_conv3d(BF16, 0, ptr(x), cin, cin, n, s, s, s, ptr(w), None, None, None, 0, ptr(out), cout, cout, ptr(part), 0, st), "conv")
def inb():
    check(lib().hdf_op_in_bwd(BF16, ptr(da), c, ptr(y), c, ptr(sc), ptr(sh), ptr(mu), ptr(rs), ptr(gamma), ptr(dy), c, ptr(dg), ptr(db), n, c, vox, ptr(ws), st), "in_bwd")
for mode in ("alone", "after 1 conv", "after 3 convs"):
    for _ in range(5):
        conv(); inb()
    torch.cuda.synchronize()
    tot = 0.0
    reps = 30
    evs = []
    for _ in range(reps):
        if mode != "alone":
            for _k in range(1 if mode == "after 1 conv" else 3): conv()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); inb(); e1.record(); evs.append((e0, e1))
    torch.cuda.synchronize()
    tot = sum(a.elapsed_time(b) for a, b in evs) / reps
    print(f"in_bwd {mode}: {tot*1e3:.0f} us")
