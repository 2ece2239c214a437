"""Micro-driver: the 1x1x1 head kernels (forward / backward) and the fused loss at the four levels of the benchmark
geometry, HIP-event medians, with the bytes each launch must move."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "h-denseformer_amd")):
    sys.path.insert(0, p)
import torch

from hdf_rt._lib import BF16, check, lib, ptr

dev = "cuda:0"
st = torch.cuda.current_stream().cuda_stream
N, ncls = 2, 4


def med(fn, reps=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]


for size, C in ((128, 32), (64, 64), (32, 128), (16, 256)):
    vox = size ** 3
    x = torch.randn(N, vox, C, device=dev).to(torch.bfloat16)
    sc, sh = torch.rand(N, C, device=dev) + 0.5, torch.randn(N, C, device=dev) * 0.1
    w, b = torch.randn(ncls, C, device=dev) * 0.1, torch.zeros(ncls, device=dev)
    logits = torch.empty(N, ncls, vox, device=dev, dtype=torch.bfloat16)
    dl = torch.randn(N, ncls, vox, device=dev).to(torch.bfloat16)
    dx = torch.zeros(N, vox, C, device=dev, dtype=torch.bfloat16)
    dw, db = torch.zeros(ncls, C, device=dev), torch.zeros(ncls, device=dev)
    t = med(lambda: check(lib().hdf_op_head_fwd(BF16, ptr(x), C, ptr(sc), ptr(sh), ptr(w), ptr(b), ptr(logits), N, C, ncls,
                                                vox, st), "head_fwd"))
    by = x.numel() * 2 + logits.numel() * 2
    print(f"head_fwd {C}ch @{size}^3: {t:7.1f} us  {by / t / 1e6:6.2f} TB/s")
    for acc in (0, 1):
        t = med(lambda: check(lib().hdf_op_head_bwd(BF16, ptr(dl), ptr(x), C, ptr(sc), ptr(sh), ptr(w), ptr(dx), C, acc,
                                                    ptr(dw), ptr(db), N, C, ncls, vox, st), "head_bwd"))
        by = x.numel() * 2 * (2 + acc) + dl.numel() * 2
        print(f"head_bwd {C}ch @{size}^3 accumulate={acc}: {t:7.1f} us  {by / t / 1e6:6.2f} TB/s")

# fused deep-supervision loss, forward and backward
outs = [torch.randn(N, ncls, 128 >> i, 128 >> i, 128 >> i, device=dev).to(torch.bfloat16) for i in range(4)]
douts = [torch.empty_like(o) for o in outs]
lab = torch.randint(0, ncls, (N, 128, 128, 128), device=dev)
onehot = torch.nn.functional.one_hot(lab, ncls).movedim(-1, 1).float().contiguous()
ws = torch.empty(lib().hdf_loss_workspace_bytes(N), dtype=torch.uint8, device=dev)
loss = torch.zeros(1, device=dev)
gup = torch.ones(1, device=dev)
t = med(lambda: check(lib().hdf_loss_forward(BF16, *[ptr(o) for o in outs], 4, ptr(onehot), N, ncls, 128, 128, 128, ptr(ws),
                                            ptr(loss), st), "loss_fwd"))
by = sum(o.numel() for o in outs) * 2 + onehot.numel() * 4
print(f"loss forward (4 scales + finalize + total): {t:7.1f} us  {by / t / 1e6:6.2f} TB/s")
t = med(lambda: check(lib().hdf_loss_backward(BF16, *[ptr(o) for o in outs], 4, ptr(onehot), N, ncls, 128, 128, 128, ptr(ws),
                                             ptr(gup), *[ptr(o) for o in douts], st), "loss_bwd"))
by = sum(o.numel() for o in outs) * 4 + onehot.numel() * 4
print(f"loss backward (4 scales): {t:7.1f} us  {by / t / 1e6:6.2f} TB/s")

# encoder tail (relu(IN) + skip, stored, max-pooled) at the three encoder levels
for size, C in ((128, 32), (64, 64), (32, 128)):
    vox = size ** 3
    y = torch.randn(N, vox, C, device=dev).to(torch.bfloat16)
    sk = torch.randn(N, vox, C, device=dev).to(torch.bfloat16)
    sc, sh = torch.rand(N, C, device=dev) + 0.5, torch.randn(N, C, device=dev) * 0.1
    ds = torch.empty_like(y)
    po = torch.empty(N, vox // 8, C, device=dev, dtype=torch.bfloat16)
    ix = torch.empty(N, vox // 8, C, device=dev, dtype=torch.uint8)
    h = size // 2
    t = med(lambda: check(lib().hdf_op_enc_tail(BF16, ptr(y), C, ptr(sc), ptr(sh), ptr(sk), C, ptr(ds), C, ptr(po), C, ptr(ix),
                                                N, C, h, h, h, st), "enc_tail"))
    by = y.numel() * 2 * 3 + po.numel() * 3
    print(f"enc_tail {C}ch @{size}^3: {t:7.1f} us  {by / t / 1e6:6.2f} TB/s")

# trilinear x2 of relu(x * scale + shift): the at_k features of the UpConv chain (at3: 64^3 -> 128^3, 32 channels)
for size, C in ((64, 32), (32, 64), (16, 128)):
    vox = size ** 3
    x = torch.randn(N, vox, C, device=dev).to(torch.bfloat16)
    sc, sh = torch.rand(N, C, device=dev) + 0.5, torch.randn(N, C, device=dev) * 0.1
    up = torch.empty(N, vox * 8, C, device=dev, dtype=torch.bfloat16)
    t = med(lambda: check(lib().hdf_op_upsample_fwd(BF16, ptr(x), C, ptr(sc), ptr(sh), ptr(up), C, N, C, size, size, size, st),
                          "upsample_fwd"))
    by = x.numel() * 2 + up.numel() * 2
    print(f"upsample_fwd {C}ch {size}^3 -> {2 * size}^3: {t:7.1f} us  {by / t / 1e6:6.2f} TB/s")

# its adjoint (the first launch of the UpConv chain's backward: d(at3) 128^3 -> 64^3)
for size, C in ((64, 32), (32, 64), (16, 128)):
    vox = size ** 3
    g = torch.randn(N, vox * 8, C, device=dev).to(torch.bfloat16)
    dx = torch.empty(N, vox, C, device=dev, dtype=torch.bfloat16)
    t = med(lambda: check(lib().hdf_op_upsample_bwd(BF16, ptr(g), C, ptr(dx), C, N, C, size, size, size, st), "upsample_bwd"))
    by = g.numel() * 2 + dx.numel() * 2
    print(f"upsample_bwd {C}ch {2 * size}^3 -> {size}^3: {t:7.1f} us  {by / t / 1e6:6.2f} TB/s")

# level-0 encoder tail with the up-sampling inside (at3 never written)
for size, C in ((128, 32),):
    vox = size ** 3
    y = torch.randn(N, vox, C, device=dev).to(torch.bfloat16)
    low = torch.randn(N, vox // 8, C, device=dev).to(torch.bfloat16)
    sc, sh = torch.rand(N, C, device=dev) + 0.5, torch.randn(N, C, device=dev) * 0.1
    ds = torch.empty_like(y)
    po = torch.empty(N, vox // 8, C, device=dev, dtype=torch.bfloat16)
    ix = torch.empty(N, vox // 8, C, device=dev, dtype=torch.uint8)
    h = size // 2
    t = med(lambda: check(lib().hdf_op_enc_tail_up(BF16, ptr(y), C, ptr(sc), ptr(sh), ptr(low), C, ptr(sc), ptr(sh), ptr(ds), C,
                                                   ptr(po), C, ptr(ix), N, C, h, h, h, st), "enc_tail_up"))
    by = y.numel() * 2 * 2 + low.numel() * 2 + po.numel() * 3
    print(f"enc_tail_up {C}ch @{size}^3: {t:7.1f} us  {by / t / 1e6:6.2f} TB/s")

# max-pool backward into the complete ds gradient + first pass of the InstanceNorm backward (the launch in front of the
# backward's fork at level 0)
for size, C in ((128, 32), (64, 64)):
    vox = size ** 3
    h = size // 2
    g = torch.randn(N, vox // 8, C, device=dev).to(torch.bfloat16)
    ix = torch.randint(0, 8, (N, vox // 8, C), device=dev, dtype=torch.uint8)
    din = torch.randn(N, vox, C, device=dev).to(torch.bfloat16)
    y = torch.randn(N, vox, C, device=dev).to(torch.bfloat16)
    sc, sh = torch.rand(N, C, device=dev) + 0.5, torch.randn(N, C, device=dev) * 0.1
    mu, rs = torch.randn(N, C, device=dev) * 0.1, torch.rand(N, C, device=dev) + 0.5
    rows = lib().hdf_op_maxpool_bwd_in_rows(C, h, h, h)
    part = torch.empty(N, rows, C, 2, device=dev)
    t = med(lambda: check(lib().hdf_op_maxpool_bwd_in(BF16, ptr(g), C, ptr(ix), ptr(din), C, ptr(y), C, ptr(sc), ptr(sh), ptr(mu),
                                                      ptr(rs), ptr(part), N, C, h, h, h, st), "maxpool_bwd_in"))
    by = din.numel() * 2 * 2 + y.numel() * 2 + g.numel() * 3
    print(f"maxpool_bwd_in {C}ch @{size}^3: {t:7.1f} us  {by / t / 1e6:6.2f} TB/s")

