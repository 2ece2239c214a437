"""The 2-D operators of the native HDenseFormer_2D path (round 6): a depth of 1 selects them at the operator ABI
(include/hdf.h: hdf_op_conv3d / hdf_op_conv3d_wgrad).  Reference: the torch 2-D ops models/HDenseFormer_2D.py composes
(Conv2d k3 p1 :148-150, ConvTranspose2d k3 s2 p1 op1 :204-214 and their autograd), on storage-rounded operands.
The packed weight panel keeps the 27-tap layout with the 2-D kernel on the centre depth plane (taps 9..17), which is what
the plan's parameter embedding produces (csrc/plan.hip "2-D embedding")."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from hdf_rt._lib import BF16, F16, F32, check, lib, ptr  # noqa: E402
from hip_util import DEV, conv3d, from_cl, pack_w, rel_err, rnd, rup, st, to_cl  # noqa: E402

TOL = {F32: 2e-5, BF16: 2e-2, F16: 3e-3}


def _mk(shape, seed):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed))


def _embed_conv(w2):
    """Conv2d weight [O, I, 3, 3] -> Conv3d weight [O, I, 3, 3, 3] with the kernel on depth tap 1"""
    w3 = torch.zeros(w2.shape[0], w2.shape[1], 3, 3, 3)
    w3[:, :, 1] = w2
    return w3


SHAPES = [(16, 16, (8, 8), 1), (32, 32, (24, 40), 2), (64, 32, (32, 32), 1), (32, 64, (72, 80), 2), (128, 96, (10, 9), 2),
          (64, 64, (96, 96), 1), (16, 48, (49, 51), 1), (256, 256, (16, 16), 2), (512, 256, (8, 8), 2)]


@pytest.mark.parametrize("dtype", [F32, BF16, F16])
@pytest.mark.parametrize("cin,cout,size,n", SHAPES)
def test_conv2d_s1(dtype, cin, cout, size, n):
    x, w, b = _mk((n, cin) + size, 1), _mk((cout, cin, 3, 3), 2) * (cin * 9) ** -0.5, _mk((cout,), 3)
    ref = F.conv2d(rnd(x, dtype), rnd(w, dtype), b, padding=1)
    wp = pack_w(_embed_conv(w), dtype, cout, cin, rup(cout, 32), cin, cin * 27, 27, 0)
    out, part = conv3d(dtype, 0, to_cl(x.unsqueeze(2), dtype), cin, wp, cout, bias=b.to(DEV), stats=True)
    torch.cuda.synchronize()
    got = from_cl(out)[:, :, 0]
    assert rel_err(got, ref) < TOL[dtype]
    tiles = part.shape[0] // n
    s = part.view(n, tiles, -1, 2).sum(1).cpu()[:, :cout]
    assert rel_err(s[..., 0], ref.sum((2, 3))) < 5e-3 + TOL[dtype]
    assert rel_err(s[..., 1], (ref * ref).sum((2, 3))) < 5e-3 + TOL[dtype]


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("cin,cout,size,n", [(32, 32, (40, 24), 2), (64, 32, (80, 72), 1), (64, 128, (16, 24), 2)])
def test_conv2d_input_transform_accumulate_and_pitch(dtype, cin, cout, size, n):
    """the producer's InstanceNorm + ReLU on load, input / output channel slices of wider buffers, out += conv"""
    x, w = _mk((n, cin) + size, 4), _mk((cout, cin, 3, 3), 5) * (cin * 9) ** -0.5
    scale, shift = _mk((n, cin), 6) * 0.5 + 1.0, _mk((n, cin), 7) * 0.3
    base = _mk((n, cout) + size, 8)
    xa = rnd(torch.relu(rnd(x, dtype) * scale[:, :, None, None] + shift[:, :, None, None]), dtype)
    ref = rnd(base, dtype) + F.conv2d(xa, rnd(w, dtype), None, padding=1)
    wp = pack_w(_embed_conv(w), dtype, cout, cin, rup(cout, 32), cin, cin * 27, 27, 0)
    xcl = to_cl(x.unsqueeze(2), dtype)
    wide_in = torch.zeros((n, 1) + size + (2 * cin,), dtype=xcl.dtype, device=DEV)
    wide_in[..., cin:] = xcl
    wide_out = torch.full((n, 1) + size + (2 * cout,), 7.0, dtype=xcl.dtype, device=DEV)
    wide_out[..., :cout] = to_cl(base.unsqueeze(2), dtype)
    vin = wide_in.view(-1)[cin:]
    sc, sh = scale.to(DEV), shift.to(DEV)
    check(lib().hdf_op_conv3d(dtype, 0, ptr(vin), 2 * cin, cin, n, 1, *size, ptr(wp), None, ptr(sc), ptr(sh), 1,
                              ptr(wide_out), 2 * cout, cout, None, 1, st()), "conv2d")
    torch.cuda.synchronize()
    assert rel_err(from_cl(wide_out[..., :cout])[:, :, 0], ref) < TOL[dtype] * 1.5
    assert float((wide_out[..., cout:].float() - 7.0).abs().max()) == 0.0


@pytest.mark.parametrize("dtype", [F32, BF16, F16])
@pytest.mark.parametrize("cin,cout,size", [(32, 16, (8, 8)), (64, 32, (24, 40)), (128, 64, (48, 48)), (256, 128, (6, 10)),
                                           (96, 48, (9, 7))])
def test_conv_transpose2d(dtype, cin, cout, size):
    """ConvTranspose2d(k3, s2, p1, op1): 4 output-parity classes, the depth axis untouched"""
    n = 2
    x, w, b = _mk((n, cin) + size, 8), _mk((cin, cout, 3, 3), 9) * (cin * 9 / 4) ** -0.5, _mk((cout,), 10)
    ref = F.conv_transpose2d(rnd(x, dtype), rnd(w, dtype), b, stride=2, padding=1, output_padding=1)
    w3 = torch.zeros(cin, cout, 3, 3, 3)
    w3[:, :, 1] = w
    wp = pack_w(w3, dtype, cout, cin, rup(cout, 32), cin, 27, cout * 27, 0)
    out, _ = conv3d(dtype, 2, to_cl(x.unsqueeze(2), dtype), cin, wp, cout, bias=b.to(DEV))
    torch.cuda.synchronize()
    assert out.shape[1] == 1
    assert rel_err(from_cl(out)[:, :, 0], ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", [F32, BF16, F16])
@pytest.mark.parametrize("cin,cout,size", [(16, 32, (16, 16)), (32, 64, (48, 80)), (64, 128, (24, 24)), (48, 96, (10, 14))])
def test_conv2d_stride2(dtype, cin, cout, size):
    """stride-2 gather conv in y and x = the data gradient of ConvTranspose2d(k3, s2, p1, op1)"""
    n = 2
    x, w = _mk((n, cin) + size, 11), _mk((cout, cin, 3, 3), 12) * (cin * 9) ** -0.5
    ref = F.conv2d(rnd(x, dtype), rnd(w, dtype), None, stride=2, padding=1)
    wp = pack_w(_embed_conv(w), dtype, cout, cin, rup(cout, 32), cin, cin * 27, 27, 0)
    out, _ = conv3d(dtype, 1, to_cl(x.unsqueeze(2), dtype), cin, wp, cout)
    torch.cuda.synchronize()
    assert rel_err(from_cl(out)[:, :, 0], ref) < TOL[dtype]


def _wgrad(dtype, stride, s_cl, sc, l_cl, lc, dims, sc_store, lc_store, s_scale=None, s_shift=None, s_relu=0):
    n = s_cl.shape[0]
    wsb = lib().hdf_op_wgrad_workspace_bytes(stride, n, 1, *dims, sc, lc)
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    dw = torch.full((sc_store, lc_store, 27), 7.0, dtype=torch.float32, device=DEV)
    check(lib().hdf_op_conv3d_wgrad(dtype, stride, ptr(s_cl), s_cl.shape[-1], sc, ptr(l_cl), l_cl.shape[-1], lc, n, 1,
                                    *dims, ptr(s_scale), ptr(s_shift), s_relu, None, None, 0, ptr(dw), sc_store, lc_store,
                                    0, ptr(ws), wsb, st()), "wgrad")
    torch.cuda.synchronize()
    return dw.cpu()


@pytest.mark.parametrize("dtype", [F32, BF16, F16])
@pytest.mark.parametrize("cin,cout,size,n", [(16, 16, (8, 8), 1), (32, 32, (24, 40), 2), (64, 48, (9, 7), 2),
                                            (32, 32, (96, 112), 2), (64, 32, (48, 50), 3), (256, 128, (16, 16), 2)])
def test_conv2d_wgrad(dtype, cin, cout, size, n):
    """dW of Conv2d: the 9 taps of the centre depth plane of the [.., 27] gradient; the other 18 are written as zeros"""
    x, dy = _mk((n, cin) + size, 13), _mk((n, cout) + size, 14)
    w = torch.zeros(cout, cin, 3, 3, requires_grad=True)
    F.conv2d(rnd(x, dtype), w, None, padding=1).backward(rnd(dy, dtype))
    got = _wgrad(dtype, 1, to_cl(dy.unsqueeze(2), dtype), cout, to_cl(x.unsqueeze(2), dtype), cin, size, cout, cin)
    got = got.view(cout, cin, 3, 3, 3)
    assert rel_err(got[:, :, 1], w.grad) < TOL[dtype]
    assert float(got[:, :, 0].abs().max()) == 0.0 and float(got[:, :, 2].abs().max()) == 0.0


@pytest.mark.parametrize("dtype", [F32, BF16, F16])
@pytest.mark.parametrize("cin,cout,size,xf", [(32, 16, (4, 4), 0), (64, 32, (24, 40), 1), (128, 64, (16, 16), 0),
                                             (96, 48, (7, 9), 1)])
def test_conv_transpose2d_wgrad(dtype, cin, cout, size, xf):
    n = 2
    osz = tuple(2 * s for s in size)
    x, dy = _mk((n, cin) + size, 17), _mk((n, cout) + osz, 18)
    sc = (torch.rand(n, cin, generator=torch.Generator().manual_seed(29)) + 0.5) if xf else None
    sh = (torch.randn(n, cin, generator=torch.Generator().manual_seed(30)) * 0.3) if xf else None
    xin = rnd(x, dtype)
    if xf:
        xin = rnd(torch.relu(xin * sc[:, :, None, None] + sh[:, :, None, None]), dtype)
    w = torch.zeros(cin, cout, 3, 3, requires_grad=True)
    F.conv_transpose2d(xin, w, None, stride=2, padding=1, output_padding=1).backward(rnd(dy, dtype))
    scd, shd = (sc.to(DEV), sh.to(DEV)) if xf else (None, None)
    got = _wgrad(dtype, 2, to_cl(x.unsqueeze(2), dtype), cin, to_cl(dy.unsqueeze(2), dtype), cout, size, cin, cout,
                 scd, shd, xf).view(cin, cout, 3, 3, 3)
    assert rel_err(got[:, :, 1], w.grad) < TOL[dtype]
    assert float(got[:, :, 0].abs().max()) == 0.0 and float(got[:, :, 2].abs().max()) == 0.0
