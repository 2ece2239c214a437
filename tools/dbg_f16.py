import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "h-denseformer_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from hdf_rt._lib import BF16, F16, F32, check, lib, ptr
from hip_util import DEV, from_cl, to_cl, st

torch.manual_seed(0)
n, c, size = 2, 32, (8, 8, 8)
x = (torch.randint(-8, 9, (n, c) + size).float() / 8)
res = {}
for dt in (BF16, F16, F32):
    gcl = to_cl(x, dt)
    hi = torch.zeros((n,) + tuple(2 * s for s in size) + (c,), dtype=gcl.dtype, device=DEV)
    one = torch.ones(n, c, device=DEV)
    zero = torch.zeros(n, c, device=DEV)
    check(lib().hdf_op_upsample_fwd(dt, ptr(gcl), c, ptr(one), ptr(zero), ptr(hi), c, n, c, *size, st()), "upf")
    lo = torch.empty((n,) + size + (c,), dtype=gcl.dtype, device=DEV)
    check(lib().hdf_op_upsample_bwd(dt, ptr(hi), c, ptr(lo), c, n, c, *size, st()), "upb")
    torch.cuda.synchronize()
    res[dt] = (from_cl(hi), from_cl(lo))
for k in (0, 1):
    a, b, f = res[BF16][k], res[F16][k], res[F32][k]
    print("tensor", k, "bf16 vs f32", float((a - f).abs().max()), "f16 vs f32", float((b - f).abs().max()))
    bad = (b - f).abs() > 0.05
    idx = bad.nonzero()
    print(" bad count", int(bad.sum()), "of", bad.numel())
    if len(idx):
        print(" first bad", idx[0].tolist(), "last bad", idx[-1].tolist())
        print(" bad per sample", [int(bad[i].sum()) for i in range(n)], "per channel-chunk", [int(bad[:, j*8:(j+1)*8].sum()) for j in range(c//8)])
