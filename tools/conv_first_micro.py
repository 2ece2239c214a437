#!/usr/bin/env python3
"""The encoder's first layer alone at the benchmark size (4 -> 32 channels, 128^3, batch 2, bf16): hdf_op_conv3d_first
(tap-packed K, csrc/conv_first.hip) against the generic stride-1 path on the 16-channel padded row (hdf_op_conv3d ->
conv_ws2_kernel<32,32>).  Prints median us and the output bandwidth (the layer's floor is its 268 MB of output)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "h-denseformer_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from hdf_rt._lib import BF16, check, lib, ptr  # noqa: E402
from hip_util import DEV, conv3d, pack_w, st  # noqa: E402

N, C, CO, S = 2, 4, 32, int(sys.argv[1]) if len(sys.argv) > 1 else 128
x = torch.randn(N, C, S, S, S)
w = torch.randn(CO, C, 3, 3, 3) * 0.2
xin = torch.zeros(N, S, S, S, 16, dtype=torch.bfloat16, device=DEV)
xin[..., :C] = x.to(DEV).permute(0, 2, 3, 4, 1).to(torch.bfloat16)
out = torch.empty(N, S, S, S, CO, dtype=torch.bfloat16, device=DEV)
part = torch.empty(N, 512, CO, 2, device=DEV)
wd = w.to(DEV)


def med(fn, reps=20):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    return sorted(ts)[len(ts) // 2]


t = med(lambda: check(lib().hdf_op_conv3d_first(BF16, ptr(xin), 16, C, N, S, S, S, ptr(wd), None, ptr(out), CO, CO, ptr(part),
                                                st()), "first"))
print(f"conv3d_first  {C}->{CO} @{S}^3 x{N}: {t:7.1f} us   {out.numel() * 2 / t / 1e6:5.2f} TB/s of output")
o1 = out.clone()
wpad = torch.zeros(CO, 16, 3, 3, 3)
wpad[:, :C] = w
wp = pack_w(wpad, BF16, CO, 16, CO, 16, 16 * 27, 27, 0)
t0 = med(lambda: conv3d(BF16, 0, xin, 16, wp, CO, stats=True, out=out, out_pitch=CO))
print(f"generic path  16->{CO} @{S}^3 x{N}: {t0:7.1f} us   {out.numel() * 2 / t0 / 1e6:5.2f} TB/s of output")
torch.cuda.synchronize()
print("max |first - generic| =", float((o1.float() - out.float()).abs().max()))
