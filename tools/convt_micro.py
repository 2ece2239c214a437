"""Micro-driver: ConvTranspose3d forward (mode 2) / its stride-2 gather dgrad (mode 1) / stride-2 weight gradient through the
C ABI, timed with HIP events (diagnostic; also the target of rocprofv3 --pmc runs)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "h-denseformer_amd")):
    sys.path.insert(0, p)
import torch

from hdf_rt._lib import BF16, check, lib, ptr

ap = argparse.ArgumentParser()
ap.add_argument("--op", default="convt", choices=["convt", "gather", "wgrad2"])
ap.add_argument("--cin", type=int, default=64)      # channels of the LOW-resolution tensor
ap.add_argument("--cout", type=int, default=32)     # channels of the HIGH-resolution tensor
ap.add_argument("--size", type=int, default=64)     # low-resolution extent
ap.add_argument("--n", type=int, default=2)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--xf", type=int, default=1)
ap.add_argument("--dense", type=int, default=0)  # convt: write a dense tensor instead of the upconv half of a concat row
a = ap.parse_args()
dev = "cuda:0"
st = torch.cuda.current_stream().cuda_stream
n, cl, ch, s = a.n, a.cin, a.cout, a.size
lo = torch.randn(n, s, s, s, cl, device=dev).to(torch.bfloat16)
hi = torch.randn(n, 2 * s, 2 * s, 2 * s, ch, device=dev).to(torch.bfloat16)
sc = (torch.rand(n, cl, device=dev) + 0.5) if a.xf else None
sh = (torch.randn(n, cl, device=dev) * 0.1) if a.xf else None
if a.op == "convt":
    w = (torch.randn(27 * ((ch + 31) // 32 * 32) * cl, device=dev) * 0.02).to(torch.bfloat16)
    opitch = ch if a.dense else 2 * ch
    out = torch.empty(n, 2 * s, 2 * s, 2 * s, opitch, device=dev, dtype=torch.bfloat16)   # the upconv half of a concat row

    def launch():
        check(lib().hdf_op_conv3d(BF16, 2, ptr(lo), cl, cl, n, s, s, s, ptr(w), None, ptr(sc), ptr(sh), 1, ptr(out), opitch,
                                  ch, None, 0, st), "convt")
elif a.op == "gather":
    w = (torch.randn(27 * ((cl + 31) // 32 * 32) * ch, device=dev) * 0.02).to(torch.bfloat16)
    out = torch.empty(n, s, s, s, cl, device=dev, dtype=torch.bfloat16)

    def launch():
        check(lib().hdf_op_conv3d(BF16, 1, ptr(hi), ch, ch, n, 2 * s, 2 * s, 2 * s, ptr(w), None, None, None, 0, ptr(out),
                                  cl, cl, None, 0, st), "gather")
else:
    wsb = lib().hdf_op_wgrad_workspace_bytes(2, n, s, s, s, cl, ch)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    dw = torch.zeros(cl, ch, 27, device=dev)

    def launch():
        check(lib().hdf_op_conv3d_wgrad(BF16, 2, ptr(lo), cl, cl, ptr(hi), ch, ch, n, s, s, s, ptr(sc), ptr(sh), 1 if a.xf else 0,
                                        None, None, 0, ptr(dw), cl, ch, 0, ptr(ws), wsb, st), "wgrad2")
flops = 2.0 * 27 * cl * ch * s ** 3 * n
for _ in range(3):
    launch()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.reps):
    launch()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.reps
print(f"{a.op} {cl}<->{ch} @{s}^3 x{n}: {ms * 1e3:.1f} us  {flops / ms / 1e9:.1f} TFLOP/s")
