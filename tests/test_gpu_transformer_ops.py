"""GPU parity of the transformer / patch-embedding / head operator entries of the C ABI (include/hdf.h hdf_op_attention_*,
hdf_op_dense_layer_*, hdf_op_block_out_*, hdf_op_patch_embed_*, hdf_op_head_*) against plain torch fp32 CPU ops -- the
primitives the reference composes in models/HDenseFormer.py:11-145,223-227 (nn.Linear, LayerNorm, exact GELU, softmax,
matmul, Dropout, Conv3d).  Token counts cover the shapes of BASELINE.json's configs: 8 (32^3), 27 (48^3, odd grid),
64 (64^3), 512 (128^3, the benchmark), 729 (144^3), 1000 (160^3); token dims 64 / 128 / 192 (n_filters 16 / 32 / 48).

Dropout uses the counter-hash masks of oracle/detgen.py (same integer recipe as hdf_common.h:hdf_keep), so train-mode
runs are compared element-wise.  Tolerance: 2e-5 relative (max-abs / max-ref) for forward and backward in fp32."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from hdf_rt._lib import BF16, F32, check, lib, ptr  # noqa: E402
from hip_util import DEV, from_cl, rel_err, st, to_cl  # noqa: E402
from oracle import detgen  # noqa: E402

TOL = 2e-5


def _g(seed):
    return torch.Generator().manual_seed(seed)


def _mask(seed, site, shape, rows_per_m, m):
    """dropout multiplier (0 or 2) for modality m's [rows_per_m, width] slab: element index = t*width + o"""
    if seed is None:
        return torch.ones(shape)
    keep = detgen.dropout_keep(seed, site, int(np.prod(shape)), 0.5)
    return torch.from_numpy(keep.reshape(shape)).float() * 2.0


def _ptr_array(tensors):
    arr = (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    return arr


# ------------------------------------------------------------------------------------------------ attention
@pytest.mark.parametrize("N", [8, 27, 64, 512, 729, 1000])
@pytest.mark.parametrize("nseq", [1, 3])
def test_attention_fwd_bwd(N, nseq):
    qkv = torch.randn(nseq * N, 96, generator=_g(N)) * 1.5
    d_ob = torch.randn(nseq * N, 32, generator=_g(N + 1))
    q = qkv.clone().requires_grad_(True)
    qq, kk, vv = [t.reshape(nseq, N, 8, 4).permute(0, 2, 1, 3) for t in q.split(32, dim=-1)]
    dots = torch.matmul(qq, kk.transpose(-1, -2)) * 0.5
    att = torch.softmax(dots, dim=-1)
    ref = torch.matmul(att, vv).permute(0, 2, 1, 3).reshape(nseq * N, 32)
    ref.backward(d_ob)
    ref_lse = torch.logsumexp(dots, dim=-1).permute(0, 2, 1).reshape(nseq * N, 8)

    gq = qkv.to(DEV)
    ob = torch.empty(nseq * N, 32, device=DEV)
    lse = torch.empty(nseq * N, 8, device=DEV)
    dq = torch.empty(nseq * N, 96, device=DEV)
    check(lib().hdf_op_attention_fwd(ptr(gq), nseq, N, ptr(ob), ptr(lse), st()), "attention_fwd")
    gd = d_ob.to(DEV)
    check(lib().hdf_op_attention_bwd(ptr(gq), ptr(ob), ptr(lse), ptr(gd), ptr(dq), nseq, N, st()), "attention_bwd")
    torch.cuda.synchronize()
    assert rel_err(ob.cpu(), ref.detach()) < TOL
    assert rel_err(lse.cpu(), ref_lse.detach()) < TOL
    assert rel_err(dq.cpu(), q.grad) < TOL


# the backward with what autocast does to the matmuls of Dense_Attention (16-bit operands, fp32 accumulate): tolerance =
# the operand rounding (2^-9 for bf16, 2^-12 for f16) against max|ref|; lse comes from the fp32 scores and stays at the fp32 tolerance
AMP_TOL = {1: 1.5e-2, 2: 2e-3}


@pytest.mark.parametrize("N", [8, 27, 64, 512, 729, 1000])
@pytest.mark.parametrize("nseq", [1, 3])
@pytest.mark.parametrize("dtype", [0, 1, 2])
def test_attention_amp_bwd(N, nseq, dtype):
    qkv = torch.randn(nseq * N, 96, generator=_g(N + 7)) * 1.5
    d_ob = torch.randn(nseq * N, 32, generator=_g(N + 8))
    q = qkv.clone().requires_grad_(True)
    qq, kk, vv = [t.reshape(nseq, N, 8, 4).permute(0, 2, 1, 3) for t in q.split(32, dim=-1)]
    dots = torch.matmul(qq, kk.transpose(-1, -2)) * 0.5
    ref = torch.matmul(torch.softmax(dots, dim=-1), vv).permute(0, 2, 1, 3).reshape(nseq * N, 32)
    ref.backward(d_ob)
    ref_lse = torch.logsumexp(dots, dim=-1).permute(0, 2, 1).reshape(nseq * N, 8)

    gq, gd = qkv.to(DEV), d_ob.to(DEV)
    ob = torch.full((nseq * N, 32), float("nan"), device=DEV)
    lse = torch.full((nseq * N, 8), float("nan"), device=DEV)
    dq = torch.full((nseq * N, 96), float("nan"), device=DEV)
    check(lib().hdf_op_attention_fwd(ptr(gq), nseq, N, ptr(ob), ptr(lse), st()), "attention_fwd")
    check(lib().hdf_op_attention_amp_bwd(dtype, ptr(gq), ptr(ob), ptr(lse), ptr(gd), ptr(dq), nseq, N, st()),
          "attention_amp_bwd")
    torch.cuda.synchronize()
    tol = AMP_TOL.get(dtype, TOL)
    assert rel_err(lse.cpu(), ref_lse.detach()) < TOL
    assert rel_err(ob.cpu(), ref.detach()) < TOL
    for i, name in enumerate("qkv"):      # per third: dq, dk, dv have different magnitudes
        assert rel_err(dq.cpu()[:, 32 * i:32 * i + 32], q.grad[:, 32 * i:32 * i + 32]) < tol, name


# ------------------------------------------------------------------------------------------------ dense layer
def _layer_params(DM, layer, M, seed):
    g = _g(seed)
    K = DM + 32 * layer
    shapes = [(32, K), (32,), (32,), (32,), (96, 32), (32, 32), (32,), (32,), (32,), (64, 32), (64,), (32, 64), (32,)]
    out = []
    for i, s in enumerate(shapes):
        fan = s[1] if len(s) == 2 else 1
        t = torch.randn((M,) + s, generator=g) * (fan ** -0.5 if len(s) == 2 else 0.1)
        if i in (2, 7):
            t = t + 1.0                      # LayerNorm weights around 1
        out.append(t)
    return out


def _ref_layer(Fin, P, m, layer, DM, B, N, seed, block):
    """one dense layer for modality m on [B*N, DMF] features (plain torch, autograd-able)"""
    K = DM + 32 * layer
    w0, b0, g1, be1, wqkv, wout, bout, g2, be2, w1, b1, w2, b2 = [p[m] for p in P]
    rows = B * N

    def mk(kind, width):
        return _mask(seed, detgen.site_id(m, block, layer, kind), (rows, width), rows, m)
    h0 = F.linear(Fin[:, :K], w0, b0)
    t = F.layer_norm(h0, (32,), g1, be1, 1e-5)
    qkv = F.linear(t, wqkv)
    qq, kk, vv = [x.reshape(B, N, 8, 4).permute(0, 2, 1, 3) for x in qkv.split(32, dim=-1)]
    att = torch.softmax(torch.matmul(qq, kk.transpose(-1, -2)) * 0.5, dim=-1)
    ob = torch.matmul(att, vv).permute(0, 2, 1, 3).reshape(rows, 32)
    h1 = F.linear(ob, wout, bout) * mk(detgen.KIND_ATTN_OUT, 32) + h0

    def ff(x, ka, kb):
        u = F.layer_norm(x, (32,), g2, be2, 1e-5)
        z = F.gelu(F.linear(u, w1, b1)) * mk(ka, 64)
        return F.linear(z, w2, b2) * mk(kb, 32)
    h2 = ff(h1, detgen.KIND_FF1_A, detgen.KIND_FF1_B) + h1
    return ff(h2, detgen.KIND_FF2_A, detgen.KIND_FF2_B)


@pytest.mark.parametrize("DM,N,B,M,layer,train", [(64, 8, 2, 2, 0, False), (64, 27, 1, 2, 3, True),
                                                  (128, 64, 2, 1, 1, True), (128, 512, 2, 2, 3, True),
                                                  (128, 512, 1, 1, 0, False), (128, 729, 1, 2, 2, False),
                                                  (192, 1000, 1, 1, 3, True), (192, 27, 2, 3, 2, False),
                                                  (256, 64, 1, 2, 3, True), (256, 27, 2, 1, 0, False)])
def test_dense_layer_fwd_bwd(DM, N, B, M, layer, train):
    DMF = DM + 128
    rows = B * N
    block, seed = 2, (4242 if train else None)
    P = _layer_params(DM, layer, M, 100 + DM + layer)
    Fin = torch.randn(M * rows, DMF, generator=_g(7))
    dFin = torch.randn(M * rows, DMF, generator=_g(8))
    K = DM + 32 * layer
    # ---- reference
    Pr = [p.clone().requires_grad_(True) for p in P]
    Fr = Fin.clone().requires_grad_(True)
    feats = []
    for m in range(M):
        feats.append(_ref_layer(Fr[m * rows:(m + 1) * rows], Pr, m, layer, DM, B, N, seed, block))
    feat = torch.cat(feats, 0)
    feat.backward(dFin[:, K:K + 32])
    # ---- device: parameters of modality m at pointer + m*mstride
    sizes = [int(np.prod(p.shape[1:])) for p in P]
    offs = np.concatenate([[0], np.cumsum([(s + 15) // 16 * 16 for s in sizes])])
    mstride = int(offs[-1])
    flat = torch.zeros(M * mstride)
    for m in range(M):
        for p, o, s in zip(P, offs, sizes):
            flat[m * mstride + o: m * mstride + o + s] = p[m].flatten()
    flat = flat.to(DEV)
    gflat = torch.zeros_like(flat)
    pp = [flat[int(o):] for o in offs[:-1]]
    gp = [gflat[int(o):] for o in offs[:-1]]
    Fd = Fin.to(DEV).contiguous()
    save = torch.zeros(M * rows * 232, device=DEV)
    tr, sd = (1, seed) if train else (0, 0)
    check(lib().hdf_op_dense_layer_fwd(M, B, N, DM, block, layer, _ptr_array(pp), mstride, ptr(Fd), ptr(save), tr, sd,
                                       st()), "dense_layer_fwd")
    torch.cuda.synchronize()
    assert rel_err(Fd[:, K:K + 32].cpu(), feat.detach()) < TOL
    assert torch.equal(Fd[:, :K].cpu(), Fin[:, :K])              # the block input columns are untouched
    dFd = dFin.to(DEV).contiguous()
    scratch = torch.zeros(M * rows * 160, device=DEV)
    check(lib().hdf_op_dense_layer_bwd(M, B, N, DM, block, layer, _ptr_array(pp), _ptr_array(gp), mstride, ptr(Fd),
                                       ptr(dFd), ptr(save), ptr(scratch), tr, sd, st()), "dense_layer_bwd")
    torch.cuda.synchronize()
    # dF[:, 0:K] += W0^T dh0 ; the reference gradient w.r.t. the input columns is exactly that increment
    got = (dFd[:, :K].cpu() - dFin[:, :K])
    assert rel_err(got, Fr.grad[:, :K]) < TOL
    names = ["w0", "b0", "ln1g", "ln1b", "wqkv", "wout", "bout", "ln2g", "ln2b", "w1", "b1", "w2", "b2"]
    gcpu = gflat.cpu()
    for i, (p, o, s) in enumerate(zip(Pr, offs, sizes)):
        for m in range(M):
            g = gcpu[m * mstride + o: m * mstride + o + s].view(p.shape[1:])
            assert rel_err(g, p.grad[m]) < 5 * TOL, (names[i], m, rel_err(g, p.grad[m]))


# ------------------------------------------------------------------------------------------------ block out layer
@pytest.mark.parametrize("DM,N,B,M,train,last,dtype", [(64, 8, 2, 2, True, False, F32), (128, 512, 2, 2, True, True, F32),
                                                       (128, 27, 1, 1, False, True, BF16), (192, 1000, 1, 1, False, False, F32),
                                                       (128, 729, 1, 2, True, True, F32), (256, 64, 2, 1, True, False, F32)])
def test_block_out_fwd_bwd(DM, N, B, M, train, last, dtype):
    DMF, rows, block = DM + 128, B * N, 1
    seed = 99 if train else None
    g = _g(DM + N)
    wa = torch.randn(M, 64, DMF, generator=g) * DMF ** -0.5
    ba = torch.randn(M, 64, generator=g) * 0.1
    wb = torch.randn(M, DM, 64, generator=g) * 0.125
    bb = torch.randn(M, DM, generator=g) * 0.1
    P = [wa, ba, wb, bb]
    Fin = torch.randn(M * rows, DMF, generator=g)
    dout = torch.randn(M * rows, DM, generator=g)
    if dtype == BF16:
        dout = dout.bfloat16().float()
    Pr = [p.clone().requires_grad_(True) for p in P]
    Fr = Fin.clone().requires_grad_(True)
    outs = []
    for m in range(M):
        x = Fr[m * rows:(m + 1) * rows]
        z = F.gelu(F.linear(x, Pr[0][m], Pr[1][m])) * _mask(seed, detgen.site_id(m, block, 4, 0), (rows, 64), rows, m)
        outs.append(F.linear(z, Pr[2][m], Pr[3][m]) * _mask(seed, detgen.site_id(m, block, 4, 1), (rows, DM), rows, m))
    out = torch.cat(outs, 0)
    out.backward(dout)
    sizes = [int(np.prod(p.shape[1:])) for p in P]
    offs = np.concatenate([[0], np.cumsum([(s + 15) // 16 * 16 for s in sizes])])
    mstride = int(offs[-1])
    flat = torch.zeros(M * mstride)
    for m in range(M):
        for p, o, s in zip(P, offs, sizes):
            flat[m * mstride + o: m * mstride + o + s] = p[m].flatten()
    flat = flat.to(DEV)
    gflat = torch.zeros_like(flat)
    pp = [flat[int(o):] for o in offs[:-1]]
    gp = [gflat[int(o):] for o in offs[:-1]]
    Fd = Fin.to(DEV)
    tr, sd = (1, seed) if train else (0, 0)
    tdt = torch.bfloat16 if dtype == BF16 else torch.float32
    if last:
        attnall = torch.zeros(B, N, M * DM, dtype=tdt, device=DEV)
        check(lib().hdf_op_block_out_fwd(M, B, N, DM, block, _ptr_array(pp), mstride, ptr(Fd), None, ptr(attnall),
                                         dtype, tr, sd, st()), "block_out_fwd")
        got = attnall.float().cpu().view(B, N, M, DM).permute(2, 0, 1, 3).reshape(M * rows, DM)
        d_att = dout.view(M, B, N, DM).permute(1, 2, 0, 3).reshape(B, N, M * DM).to(tdt).to(DEV).contiguous()
    else:
        nxt = torch.zeros(M * rows, DMF, device=DEV)
        check(lib().hdf_op_block_out_fwd(M, B, N, DM, block, _ptr_array(pp), mstride, ptr(Fd), ptr(nxt), None, dtype, tr,
                                         sd, st()), "block_out_fwd")
        got = nxt[:, :DM].cpu()
        dnext = torch.zeros(M * rows, DMF)
        dnext[:, :DM] = dout
        dnext = dnext.to(DEV)
    torch.cuda.synchronize()
    assert rel_err(got, out.detach()) < (1e-2 if dtype == BF16 else TOL)
    dF = torch.zeros(M * rows, DMF, device=DEV)
    if last:
        check(lib().hdf_op_block_out_bwd(M, B, N, DM, block, _ptr_array(pp), _ptr_array(gp), mstride, ptr(Fd), None,
                                         ptr(d_att), dtype, ptr(dF), tr, sd, st()), "block_out_bwd")
    else:
        check(lib().hdf_op_block_out_bwd(M, B, N, DM, block, _ptr_array(pp), _ptr_array(gp), mstride, ptr(Fd), ptr(dnext),
                                         None, dtype, ptr(dF), tr, sd, st()), "block_out_bwd")
    torch.cuda.synchronize()
    assert rel_err(dF.cpu(), Fr.grad) < TOL
    gcpu = gflat.cpu()
    for i, (p, o, s) in enumerate(zip(Pr, offs, sizes)):
        for m in range(M):
            gg = gcpu[m * mstride + o: m * mstride + o + s].view(p.shape[1:])
            assert rel_err(gg, p.grad[m]) < 5 * TOL, (i, m)


# ------------------------------------------------------------------------------------------------ patch embedding
@pytest.mark.parametrize("DM,size,B,M,train", [(64, (32, 32, 32), 2, 2, False), (128, (48, 32, 64), 1, 3, True),
                                               (192, (32, 48, 32), 2, 1, True), (128, (128, 128, 128), 1, 2, False),
                                               (256, (32, 32, 48), 1, 2, True)])
def test_patch_embed_fwd_bwd(DM, size, B, M, train):
    D, H, W = size
    N = (D // 16) * (H // 16) * (W // 16)
    DMF, rows = DM + 128, B * N
    seed = 31337 if train else None
    g = _g(DM)
    x = torch.rand(B, M, D, H, W, generator=g)
    w = torch.randn(M, DM, 1, 16, 16, 16, generator=g) * 4096 ** -0.5
    b = torch.randn(M, DM, generator=g) * 0.1
    pos = torch.randn(M, 1, N, DM, generator=g) * 0.1
    dF = torch.randn(M * rows, DMF, generator=g)
    wr, br, pr = [t.clone().requires_grad_(True) for t in (w, b, pos)]
    toks = []
    for m in range(M):
        t = F.conv3d(x[:, m:m + 1], wr[m], br[m], stride=16).flatten(2).transpose(1, 2) + pr[m]
        toks.append((t * _mask(seed, detgen.site_emb(m), (B, N, DM), rows, m)).reshape(rows, DM))
    tok = torch.cat(toks, 0)
    tok.backward(dF[:, :DM])
    sizes = [N * DM, DM * 4096, DM]
    offs = np.concatenate([[0], np.cumsum([(s + 15) // 16 * 16 for s in sizes])])
    mstride = int(offs[-1])
    flat = torch.zeros(M * mstride)
    for m in range(M):
        for t, o, s in zip((pos, w, b), offs, sizes):
            flat[m * mstride + o: m * mstride + o + s] = t[m].flatten()
    flat = flat.to(DEV)
    gflat = torch.zeros_like(flat)
    xd = x.to(DEV).contiguous()
    Fd = torch.zeros(M * rows, DMF, device=DEV)
    tr, sd = (1, seed) if train else (0, 0)
    check(lib().hdf_op_patch_embed_fwd(ptr(xd), M, B, D, H, W, DM, ptr(flat[int(offs[1]):]), ptr(flat[int(offs[2]):]),
                                       ptr(flat), mstride, ptr(Fd), tr, sd, st()), "patch_embed_fwd")
    torch.cuda.synchronize()
    assert rel_err(Fd[:, :DM].cpu(), tok.detach()) < TOL
    dFd = dF.to(DEV)
    scratch = torch.zeros(M * rows * DM, device=DEV)
    check(lib().hdf_op_patch_embed_bwd(ptr(xd), M, B, D, H, W, DM, ptr(dFd), mstride, ptr(gflat[int(offs[1]):]),
                                       ptr(gflat[int(offs[2]):]), ptr(gflat), ptr(scratch), tr, sd, st()),
          "patch_embed_bwd")
    torch.cuda.synchronize()
    gcpu = gflat.cpu()
    for m in range(M):
        for name, t, o, s in zip(("pos", "w", "b"), (pr, wr, br), offs, sizes):
            gg = gcpu[m * mstride + o: m * mstride + o + s].view(t.shape[1:])
            assert rel_err(gg, t.grad[m]) < 5 * TOL, (name, m, rel_err(gg, t.grad[m]))


# ------------------------------------------------------------------------------------------------ 1x1x1 heads
@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("C_,ncls,size,n,xf", [(32, 4, (16, 16, 16), 2, True), (64, 3, (8, 12, 10), 1, False),
                                              (48, 2, (9, 7, 5), 2, True), (256, 4, (4, 4, 4), 2, False),
                                              (384, 3, (2, 3, 4), 1, True), (512, 3, (4, 4, 4), 1, False),
                                              (32, 6, (8, 8, 12), 2, True), (64, 8, (6, 5, 7), 1, False),
                                              # vox = 2^21 (the 128^3 level): 2048 voxels per workgroup; with 8 class slots the
                                              # forward needs 72 KiB of dynamic LDS (allowed per kernel)
                                              (16, 6, (128, 128, 128), 1, True)])
def test_head_fwd_bwd(dtype, C_, ncls, size, n, xf):
    g = _g(C_ + ncls)
    tdt = torch.bfloat16 if dtype == BF16 else torch.float32
    x = torch.randn((n, C_) + size, generator=g)
    w = torch.randn(ncls, C_, 1, 1, 1, generator=g) * C_ ** -0.5
    b = torch.randn(ncls, generator=g) * 0.1
    scale = torch.rand(n, C_, generator=g) + 0.5
    shift = torch.randn(n, C_, generator=g) * 0.3
    dl = torch.randn((n, ncls) + size, generator=g).to(tdt).float()
    xs = x.to(tdt).float()
    act = (F.relu(xs * scale[:, :, None, None, None] + shift[:, :, None, None, None]) if xf else xs).requires_grad_(True)
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.conv3d(act, wr, br)
    ref.backward(dl)
    vox = int(np.prod(size))
    xcl = to_cl(x, dtype)
    logits = torch.empty((n, ncls) + size, dtype=tdt, device=DEV)
    sc, sh = (scale.to(DEV), shift.to(DEV)) if xf else (None, None)
    wd, bd = w.to(DEV).contiguous(), b.to(DEV)
    check(lib().hdf_op_head_fwd(dtype, ptr(xcl), C_, ptr(sc), ptr(sh), ptr(wd), ptr(bd), ptr(logits), n, C_, ncls, vox,
                                st()), "head_fwd")
    tol = TOL if dtype == F32 else 1e-2
    torch.cuda.synchronize()
    assert rel_err(logits.float().cpu(), ref.detach()) < tol
    dx = torch.zeros((n,) + size + (C_,), dtype=tdt, device=DEV)
    dw = torch.zeros(ncls, C_, device=DEV)
    db = torch.zeros(ncls, device=DEV)
    dld = dl.to(tdt).to(DEV).contiguous()
    check(lib().hdf_op_head_bwd(dtype, ptr(dld), ptr(xcl), C_, ptr(sc), ptr(sh), ptr(wd), ptr(dx), C_, 0, ptr(dw),
                                ptr(db), n, C_, ncls, vox, st()), "head_bwd")
    torch.cuda.synchronize()
    assert rel_err(from_cl(dx), act.grad) < tol
    assert rel_err(dw.cpu(), wr.grad.view(ncls, C_)) < 5 * tol
    assert rel_err(db.cpu(), br.grad) < 5 * tol
