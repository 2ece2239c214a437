"""Diagnostic: patch-embedding forward of the HIP kernel against a float64 reference."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "h-denseformer_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
import torch.nn.functional as F
from hdf_rt._lib import check, lib, ptr

DEV = "cuda:0"
st = torch.cuda.current_stream().cuda_stream
for (DM, size, B, M) in [(16, (32, 32, 32), 2, 2), (128, (128, 128, 128), 2, 4), (64, (32, 32, 32), 2, 2)]:
    if DM % 32:
        continue
    D, H, W = size
    N = (D // 16) * (H // 16) * (W // 16)
    g = torch.Generator().manual_seed(5)
    x = torch.rand(B, M, D, H, W, generator=g)
    w = torch.randn(M, DM, 1, 16, 16, 16, generator=g) * 4096 ** -0.5
    b = torch.randn(M, DM, generator=g) * 0.1
    pos = torch.randn(M, 1, N, DM, generator=g) * 0.1
    ref = torch.cat([(F.conv3d(x[:, m:m + 1].double(), w[m].double(), b[m].double(), stride=16).flatten(2).transpose(1, 2)
                      + pos[m].double()).reshape(B * N, DM) for m in range(M)], 0)
    ref32 = torch.cat([(F.conv3d(x[:, m:m + 1], w[m], b[m], stride=16).flatten(2).transpose(1, 2)
                        + pos[m]).reshape(B * N, DM) for m in range(M)], 0)
    sizes = [N * DM, DM * 4096, DM]
    offs = np.concatenate([[0], np.cumsum([(s + 15) // 16 * 16 for s in sizes])])
    ms = int(offs[-1])
    flat = torch.zeros(M * ms)
    for m in range(M):
        for t, o, s in zip((pos, w, b), offs, sizes):
            flat[m * ms + o: m * ms + o + s] = t[m].flatten()
    flat = flat.to(DEV)
    Fd = torch.zeros(M * B * N, DM + 128, device=DEV)
    check(lib().hdf_op_patch_embed_fwd(ptr(x.to(DEV).contiguous()), M, B, D, H, W, DM, ptr(flat[int(offs[1]):]),
                                       ptr(flat[int(offs[2]):]), ptr(flat), ms, ptr(Fd), 0, 0, st), "pe")
    torch.cuda.synchronize()
    got = Fd[:, :DM].cpu().double()
    e = lambda a: float((a - ref).norm() / ref.norm())
    print(f"DM {DM} {size}: HIP vs f64 {e(got):.3e}   torch-f32 vs f64 {e(ref32.double()):.3e}")
