"""Diagnostic: the dominant conv kernel (64->32 @128^3, batch 2) launched N times on one stream against N/2 + N/2 times on
two streams: is there slack (tails, stalls) that a second resident kernel can fill?"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "h-denseformer_amd")):
    sys.path.insert(0, p)
import torch
from hdf_rt._lib import BF16, check, lib, ptr

dev = "cuda:0"
n, s, ci, co = 2, 128, 64, 32
xs = [torch.randn(n, s, s, s, ci, device=dev).to(torch.bfloat16) for _ in range(2)]
outs = [torch.empty(n, s, s, s, co, device=dev, dtype=torch.bfloat16) for _ in range(2)]
w = (torch.randn(27 * co * ci, device=dev) * 0.02).to(torch.bfloat16)
st = [torch.cuda.Stream(), torch.cuda.Stream()]


def conv(k, stream):
    check(lib().hdf_op_conv3d(BF16, 0, ptr(xs[k]), ci, ci, n, s, s, s, ptr(w), None, None, None, 0, ptr(outs[k]), co, co, None, 0,
                              stream.cuda_stream), "conv")


def timed(fn):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


N = 40
one = lambda: [conv(i & 1, st[0]) for i in range(N)]
two = lambda: [conv(i & 1, st[i & 1]) for i in range(N)]
for _ in range(2):
    one(), two()
t1, t2 = timed(one), timed(two)
print(f"{N} launches: one stream {t1 / N * 1e3:.1f} us per launch, two streams {t2 / N * 1e3:.1f} us per launch")
