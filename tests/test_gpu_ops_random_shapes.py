"""SURVEY 5.2 / VERDICT r03 #7b: randomized shapes, pitches and channel counts (hypothesis) for the operator-level C
entries -- hdf_op_conv3d (stride 1, with and without the input transform, channel slices of wider buffers),
hdf_op_conv3d_wgrad, hdf_op_maxpool_fwd/bwd, hdf_op_upsample_fwd/bwd and hdf_op_head_fwd/bwd -- against plain torch fp32
on the storage-rounded operands.  The parametrized tests of test_gpu_ops.py pin the shapes the plan uses; these look for
the ragged extent, the odd pitch or the channel count nobody thought of.  Derandomized (a fixed example database would
not travel to the GPU box): the same examples every run, so a failure is reproducible."""
import os
import sys

import pytest
import torch
import torch.nn.functional as F
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as hs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "h-denseformer_amd"), os.path.dirname(os.path.abspath(__file__))):
    sys.path.insert(0, p)
from hdf_rt._lib import BF16, F16, F32, check, lib, ptr  # noqa: E402
from hip_util import DEV, TDT, from_cl, pack_w, rel_err, rnd, rup, st  # noqa: E402

pytestmark = pytest.mark.gpu
TOL = {F32: 2e-5, BF16: 2e-2, F16: 3e-3}
CFG = dict(max_examples=24, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))

dims = hs.tuples(hs.integers(1, 5), hs.integers(1, 6), hs.integers(1, 7)).map(lambda t: (2 * t[0], 2 * t[1], 2 * t[2]))
ragged = hs.tuples(hs.integers(2, 11), hs.integers(2, 13), hs.integers(2, 17))
dtypes = hs.sampled_from([F32, BF16, F16])


def _mk(shape, seed):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed))


def _wide(x, dtype, pitch, lead):
    """channels-last storage of x [N,C,D,H,W] as a channel slice [lead, lead + C) of a buffer with `pitch` channels;
    returns (whole buffer, flat view starting at the slice)"""
    n, c = x.shape[:2]
    buf = torch.full((n,) + tuple(x.shape[2:]) + (pitch,), 3.0, dtype=TDT[dtype], device=DEV)
    buf[..., lead:lead + c] = x.to(DEV).permute(0, 2, 3, 4, 1).to(TDT[dtype])
    return buf, buf.view(-1)[lead:]


@settings(**CFG)
@given(dtype=dtypes, size=ragged, n=hs.integers(1, 3), cin16=hs.integers(1, 5), cout=hs.integers(1, 72),
       in_extra=hs.sampled_from([0, 8, 16, 40]), out_extra=hs.sampled_from([0, 1, 8, 24]), xf=hs.booleans(),
       seed=hs.integers(0, 10 ** 6))
def test_conv3d_stride1_random(dtype, size, n, cin16, cout, in_extra, out_extra, xf, seed):
    cin = 16 * cin16
    x, w, b = _mk((n, cin) + size, seed), _mk((cout, cin, 3, 3, 3), seed + 1) * (cin * 27) ** -0.5, _mk((cout,), seed + 2)
    scale, shift = _mk((n, cin), seed + 3) * 0.5 + 1.0, _mk((n, cin), seed + 4) * 0.3
    xa = rnd(x, dtype)
    if xf:
        xa = rnd(torch.relu(xa * scale[:, :, None, None, None] + shift[:, :, None, None, None]), dtype)
    ref = F.conv3d(xa, rnd(w, dtype), b, padding=1)
    wp = pack_w(w, dtype, cout, cin, rup(cout, 32), cin, cin * 27, 27, 0)
    ipitch, opitch = cin + in_extra, cout + out_extra
    # 16-byte aligned input view: the leading channels of the wide buffer are a multiple of 8 (16-bit) / 4 (fp32) elements
    ibuf, iview = _wide(x, dtype, ipitch, in_extra)
    obuf = torch.full((n,) + size + (opitch,), 7.0, dtype=TDT[dtype], device=DEV)
    sc, sh = (scale.to(DEV), shift.to(DEV)) if xf else (None, None)
    bd = b.to(DEV)
    check(lib().hdf_op_conv3d(dtype, 0, ptr(iview), ipitch, cin, n, *size, ptr(wp), ptr(bd), ptr(sc), ptr(sh), 1,
                              ptr(obuf), opitch, cout, None, 0, st()), "conv")
    torch.cuda.synchronize()
    assert rel_err(from_cl(obuf[..., :cout]), ref) < TOL[dtype]
    if out_extra:
        assert float((obuf[..., cout:].float() - 7.0).abs().max()) == 0.0      # the rest of the row is untouched


@settings(**CFG)
@given(dtype=dtypes, size=ragged, n=hs.integers(1, 3), cin16=hs.integers(1, 4), cout16=hs.integers(1, 4),
       seed=hs.integers(0, 10 ** 6))
def test_conv3d_wgrad_random(dtype, size, n, cin16, cout16, seed):
    cin, cout = 16 * cin16, 16 * cout16
    x, dy = _mk((n, cin) + size, seed), _mk((n, cout) + size, seed + 1)
    w = torch.zeros(cout, cin, 3, 3, 3, requires_grad=True)
    F.conv3d(rnd(x, dtype), w, None, padding=1).backward(rnd(dy, dtype))
    dy_cl = dy.to(DEV).permute(0, 2, 3, 4, 1).to(TDT[dtype]).contiguous()
    x_cl = x.to(DEV).permute(0, 2, 3, 4, 1).to(TDT[dtype]).contiguous()
    wsb = lib().hdf_op_wgrad_workspace_bytes(1, n, *size, cout, cin)
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    dw = torch.zeros((cout, cin, 27), dtype=torch.float32, device=DEV)
    check(lib().hdf_op_conv3d_wgrad(dtype, 1, ptr(dy_cl), cout, cout, ptr(x_cl), cin, cin, n, *size, None, None, 0, None,
                                    None, 0, ptr(dw), cout, cin, 0, ptr(ws), wsb, st()), "wgrad")
    torch.cuda.synchronize()
    assert rel_err(dw.cpu().view(cout, cin, 3, 3, 3), w.grad) < TOL[dtype] * 1.5


@settings(**CFG)
@given(dtype=dtypes, size=dims, n=hs.integers(1, 3), c16=hs.integers(1, 6), seed=hs.integers(0, 10 ** 6))
def test_maxpool_and_upsample_random(dtype, size, n, c16, seed):
    c = 16 * c16
    x = _mk((n, c) + size, seed)
    x[:, :, :1] = 0.25                                  # exact ties: the FIRST maximum wins (torch's rule)
    xr = rnd(x, dtype).requires_grad_(True)
    x_cl = x.to(DEV).permute(0, 2, 3, 4, 1).to(TDT[dtype]).contiguous()
    ps = tuple(s // 2 for s in size)
    po = torch.empty((n,) + ps + (c,), dtype=x_cl.dtype, device=DEV)
    idx = torch.empty(po.shape, dtype=torch.uint8, device=DEV)
    check(lib().hdf_op_maxpool_fwd(dtype, ptr(x_cl), c, ptr(po), c, ptr(idx), n, c, *ps, st()), "pool")
    ref = F.max_pool3d(xr, 2)
    g = _mk(tuple(ref.shape), seed + 1)
    ref.backward(rnd(g, dtype))
    din = torch.zeros_like(x_cl)
    gcl = g.to(DEV).permute(0, 2, 3, 4, 1).to(TDT[dtype]).contiguous()
    check(lib().hdf_op_maxpool_bwd(dtype, ptr(gcl), c, ptr(idx), ptr(din), c, n, c, *ps, 0, st()), "poolb")
    torch.cuda.synchronize()
    assert rel_err(from_cl(po), ref.detach()) == 0
    assert rel_err(from_cl(din), xr.grad) < 1e-6
    # trilinear x2 of relu(x * scale + shift)
    scale, shift = _mk((n, c), seed + 2) * 0.5 + 1.0, _mk((n, c), seed + 3) * 0.3
    xa = torch.relu(rnd(x, dtype) * scale[:, :, None, None, None] + shift[:, :, None, None, None]).detach().requires_grad_(True)
    ref = F.interpolate(xa, scale_factor=2, mode="trilinear", align_corners=False)
    up = torch.empty((n,) + tuple(2 * s for s in size) + (c,), dtype=x_cl.dtype, device=DEV)
    sc, sh = scale.to(DEV), shift.to(DEV)
    check(lib().hdf_op_upsample_fwd(dtype, ptr(x_cl), c, ptr(sc), ptr(sh), ptr(up), c, n, c, *size, st()), "up")
    g2 = _mk(tuple(ref.shape), seed + 4)
    ref.backward(rnd(g2, dtype))
    dlo = torch.empty_like(x_cl)
    g2cl = g2.to(DEV).permute(0, 2, 3, 4, 1).to(TDT[dtype]).contiguous()
    check(lib().hdf_op_upsample_bwd(dtype, ptr(g2cl), c, ptr(dlo), c, n, c, *size, st()), "upb")
    torch.cuda.synchronize()
    assert rel_err(from_cl(up), ref.detach()) < TOL[dtype]
    assert rel_err(from_cl(dlo), xa.grad) < TOL[dtype]


@settings(**CFG)
@given(dtype=dtypes, size=dims, n=hs.integers(1, 3), c16=hs.integers(1, 6), extra=hs.integers(0, 2),
       seed=hs.integers(0, 10 ** 6))
def test_encoder_tail_with_upsampling_inside_random(dtype, size, n, c16, extra, seed):
    """hdf_op_enc_tail_up with every operand a channel slice of a wider buffer (the plan's ds is a slice of cat)"""
    c = 16 * c16
    pitch, lead = c + 16 * extra, 16 * (extra // 2)
    lo = tuple(v // 2 for v in size)
    y, low = _mk((n, c) + size, seed), _mk((n, c) + lo, seed + 1)
    scale, shift = _mk((n, c), seed + 2) * 0.5 + 1.0, _mk((n, c), seed + 3) * 0.3
    lscale, lshift = _mk((n, c), seed + 4) * 0.5 + 1.0, _mk((n, c), seed + 5) * 0.3
    bc = lambda v: v[:, :, None, None, None]
    up = F.interpolate(torch.relu(rnd(low, dtype) * bc(lscale) + bc(lshift)), scale_factor=2, mode="trilinear",
                       align_corners=False)
    ds_ref = rnd(torch.relu(rnd(y, dtype) * bc(scale) + bc(shift)) + up, dtype)
    ybuf, yv = _wide(y, dtype, pitch, lead)
    lbuf, lv = _wide(low, dtype, pitch, lead)
    dbuf, dv = _wide(torch.zeros_like(y), dtype, pitch, lead)
    po = torch.empty((n,) + lo + (c,), dtype=TDT[dtype], device=DEV)
    idx = torch.empty(po.shape, dtype=torch.uint8, device=DEV)
    dev = [t.to(DEV).contiguous() for t in (scale, shift, lscale, lshift)]
    check(lib().hdf_op_enc_tail_up(dtype, ptr(yv), pitch, ptr(dev[0]), ptr(dev[1]), ptr(lv), pitch, ptr(dev[2]), ptr(dev[3]),
                                   ptr(dv), pitch, ptr(po), c, ptr(idx), n, c, *lo, st()), "enc_tail_up")
    torch.cuda.synchronize()
    got = dbuf[..., lead:lead + c].permute(0, 4, 1, 2, 3).float().cpu()
    assert rel_err(got, ds_ref) < TOL[dtype]
    po2, _ = F.max_pool3d(got, 2, return_indices=True)
    assert bool((from_cl(po) == po2).all())
    if pitch > c:   # the channels outside the slice are untouched
        rest = torch.cat([dbuf[..., :lead], dbuf[..., lead + c:]], -1)
        assert bool((rest == 3.0).all())


@settings(**CFG)
@given(dtype=hs.sampled_from([F32, BF16]), size=ragged, n=hs.integers(1, 2), c16=hs.integers(1, 8), ncls=hs.integers(2, 8),
       xf=hs.booleans(), seed=hs.integers(0, 10 ** 6))
def test_head_random(dtype, size, n, c16, ncls, xf, seed):
    c = 16 * c16
    tdt = TDT[dtype]
    x = _mk((n, c) + size, seed)
    w = _mk((ncls, c, 1, 1, 1), seed + 1) * c ** -0.5
    b = _mk((ncls,), seed + 2) * 0.1
    scale, shift = torch.rand(n, c, generator=torch.Generator().manual_seed(seed + 3)) + 0.5, _mk((n, c), seed + 4) * 0.3
    dl = _mk((n, ncls) + size, seed + 5).to(tdt).float()
    xs = x.to(tdt).float()
    act = (F.relu(xs * scale[:, :, None, None, None] + shift[:, :, None, None, None]) if xf else xs).requires_grad_(True)
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.conv3d(act, wr, br)
    ref.backward(dl)
    vox = size[0] * size[1] * size[2]
    xcl = x.to(DEV).permute(0, 2, 3, 4, 1).to(tdt).contiguous()
    logits = torch.empty((n, ncls) + size, dtype=tdt, device=DEV)
    sc, sh = (scale.to(DEV), shift.to(DEV)) if xf else (None, None)
    wd, bd = w.to(DEV).contiguous(), b.to(DEV)
    check(lib().hdf_op_head_fwd(dtype, ptr(xcl), c, ptr(sc), ptr(sh), ptr(wd), ptr(bd), ptr(logits), n, c, ncls, vox, st()),
          "head_fwd")
    tol = 1e-4 if dtype == F32 else 1e-2
    torch.cuda.synchronize()
    assert rel_err(logits.float().cpu(), ref.detach()) < tol
    dx = torch.zeros((n,) + size + (c,), dtype=tdt, device=DEV)
    dw = torch.zeros(ncls, c, device=DEV)
    db = torch.zeros(ncls, device=DEV)
    dld = dl.to(tdt).to(DEV).contiguous()
    check(lib().hdf_op_head_bwd(dtype, ptr(dld), ptr(xcl), c, ptr(sc), ptr(sh), ptr(wd), ptr(dx), c, 0, ptr(dw), ptr(db), n,
                                c, ncls, vox, st()), "head_bwd")
    torch.cuda.synchronize()
    assert rel_err(from_cl(dx), act.grad) < tol
    assert rel_err(dw.cpu(), wr.grad.view(ncls, c)) < 5 * tol
    assert rel_err(db.cpu(), br.grad) < 5 * tol
