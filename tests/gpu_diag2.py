"""Ad-hoc: who is inexact in conv backward-input -- torch CPU fp32 or the HIP dgrad? (fp64 referee)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "h-denseformer_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import torch.nn.functional as F

from hdf_rt._lib import F32
from hip_util import conv3d, from_cl, pack_w, rup, to_cl


def rl2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def main():
    print("threads", torch.get_num_threads(), "mkldnn", torch.backends.mkldnn.is_available())
    for (cin, cout, size, n) in [(32, 16, (32, 32, 32), 2), (32, 32, (16, 16, 16), 2), (64, 64, (8, 8, 8), 2)]:
        g = torch.Generator().manual_seed(0)
        x = torch.randn((n, cin) + size, generator=g, requires_grad=True)
        w = torch.randn((cout, cin, 3, 3, 3), generator=g) * (cin * 27) ** -0.5
        dy = torch.randn((n, cout) + size, generator=g)
        y = F.conv3d(x, w, None, padding=1)
        y.backward(dy)
        x64 = x.detach().double().requires_grad_(True)
        F.conv3d(x64, w.double(), None, padding=1).backward(dy.double())
        wp = pack_w(w, F32, cin, cout, rup(cin, 32), cout, 27, cin * 27, 1)
        out, _ = conv3d(F32, 0, to_cl(dy, F32), cout, wp, cin)
        torch.cuda.synchronize()
        mine = from_cl(out)
        yg, _ = conv3d(F32, 0, to_cl(x.detach(), F32), cin, pack_w(w, F32, cout, cin, rup(cout, 32), cin, cin * 27, 27, 0), cout)
        torch.cuda.synchronize()
        y64 = F.conv3d(x.detach().double(), w.double(), None, padding=1)
        print(f"cin={cin} cout={cout} {size}: fwd torch32-vs-64 {rl2(y.detach(), y64):.2e}  hip-vs-64 {rl2(from_cl(yg), y64):.2e} | "
              f"bwd-input torch32-vs-64 {rl2(x.grad, x64.grad):.2e}  hip-vs-64 {rl2(mine, x64.grad):.2e}")
    with torch.backends.mkldnn.flags(enabled=False):
        x = torch.randn(2, 32, 16, 16, 16, requires_grad=True)
        w = torch.randn(32, 32, 3, 3, 3) * 0.03
        dy = torch.randn(2, 32, 16, 16, 16)
        F.conv3d(x, w, None, padding=1).backward(dy)
        x64 = x.detach().double().requires_grad_(True)
        F.conv3d(x64, w.double(), None, padding=1).backward(dy.double())
        print("mkldnn disabled: bwd-input torch32-vs-64", rl2(x.grad, x64.grad))


if __name__ == "__main__":
    main()
