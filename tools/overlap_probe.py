"""Diagnostic: does an HBM-bound elementwise kernel overlap with a power-capped MFMA kernel when both are in flight on two
streams?  (wgrad 64->32 @128^3 through the C ABI on one stream, a torch elementwise pass over two 268 MB tensors on the other.)"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "h-denseformer_amd")):
    sys.path.insert(0, p)
import torch
from hdf_rt._lib import BF16, check, lib, ptr

dev = "cuda:0"
n, s, sc, lc = 2, 128, 32, 64
dy = torch.randn(n, s, s, s, sc, device=dev).to(torch.bfloat16)
x = torch.randn(n, s, s, s, lc, device=dev).to(torch.bfloat16)
wsb = lib().hdf_op_wgrad_workspace_bytes(1, n, s, s, s, sc, lc)
ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
dw = torch.zeros(sc, lc, 27, device=dev)
a = torch.randn(n, s, s, s, 32, device=dev).to(torch.bfloat16)
b = torch.randn(n, s, s, s, 32, device=dev).to(torch.bfloat16)
c = torch.empty_like(a)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
REP = 10


def wgrad(stream):
    check(lib().hdf_op_conv3d_wgrad(BF16, 1, ptr(dy), sc, sc, ptr(x), lc, lc, n, s, s, s, None, None, 0, None, None, 0, ptr(dw),
                                    sc, lc, 0, ptr(ws), wsb, stream.cuda_stream), "wgrad")


def ew(k):
    for _ in range(k):
        torch.add(a, b, out=c)


def timed(fn):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


def run_w():
    for _ in range(REP):
        wgrad(s1)


def run_e():
    with torch.cuda.stream(s2):
        ew(3 * REP)


def run_both():
    for _ in range(REP):
        wgrad(s1)
        with torch.cuda.stream(s2):
            ew(3)


for _ in range(2):
    run_w(), run_e(), run_both()
tw, te, tb = timed(run_w), timed(run_e), timed(run_both)
print(f"wgrad alone {tw / REP * 1e3:.0f} us/launch; elementwise alone {te / REP * 1e3:.0f} us per 3 passes; "
      f"both on two streams {tb / REP * 1e3:.0f} us per (launch + 3 passes); sum of the two {(tw + te) / REP * 1e3:.0f} us")
