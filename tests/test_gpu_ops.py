"""GPU parity of the operator-level C ABI (include/hdf.h hdf_op_*) against plain torch fp32 CPU ops --
the same primitives the reference composes (HDenseFormer.py:148-175,199-227).
Tolerances: fp32 storage 2e-5 relative (max-abs / max-ref); bf16 storage 2e-2 against the reference
evaluated on bf16-rounded inputs (fp32 accumulate, bf16 output rounding); float16 storage 3e-3 likewise."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from hdf_rt._lib import BF16, F16, F32, check, lib, ptr  # noqa: E402
from hip_util import DEV, TDT, conv3d, from_cl, pack_w, rel_err, rnd, rup, st, to_cl  # noqa: E402

TOL = {F32: 2e-5, BF16: 2e-2, F16: 3e-3}      # storage rounding: 2^-8 bf16, 2^-11 f16


def _mk(shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g)


@pytest.mark.parametrize("dtype", [F32, BF16, F16])
@pytest.mark.parametrize("cin,cout,size,n", [(16, 16, (8, 8, 8), 1), (32, 32, (12, 16, 24), 2),
                                            (64, 32, (32, 32, 32), 1), (32, 64, (36, 36, 40), 1),
                                            (128, 96, (6, 10, 9), 2),
                                            # >= 48^3: the weights-stationary persistent kernel (conv_ws2_kernel)
                                            (16, 32, (48, 48, 48), 1), (32, 32, (48, 56, 64), 2),
                                            (64, 32, (48, 48, 50), 1), (32, 64, (52, 48, 48), 1),
                                            (16, 48, (49, 51, 53), 1)])
def test_conv3d_s1(dtype, cin, cout, size, n):
    x, w, b = _mk((n, cin) + size, 1), _mk((cout, cin, 3, 3, 3), 2) * (cin * 27) ** -0.5, _mk((cout,), 3)
    ref = F.conv3d(rnd(x, dtype), rnd(w, dtype), b, padding=1)
    wp = pack_w(w, dtype, cout, cin, rup(cout, 32), cin, cin * 27, 27, 0)
    out, part = conv3d(dtype, 0, to_cl(x, dtype), cin, wp, cout, bias=b.to(DEV), stats=True)
    torch.cuda.synchronize()
    got = from_cl(out)
    assert rel_err(got, ref) < TOL[dtype]
    # statistics partials: per (n, c) sum and sum of squares of the stored output
    tiles = part.shape[0] // n
    s = part.view(n, tiles, -1, 2).sum(1).cpu()[:, :cout]
    assert rel_err(s[..., 0], ref.sum((2, 3, 4))) < 5e-3 + TOL[dtype]
    assert rel_err(s[..., 1], (ref * ref).sum((2, 3, 4))) < 5e-3 + TOL[dtype]


def _conv3d_wr(dtype, x_cl, cin, w_packed, cout, bias=None, scale=None, shift=None, relu=0, out=None, out_pitch=None,
               accumulate=0):
    """hdf_op_conv3d_wr: the weights-in-registers kernel whatever the plan's routing rule says."""
    n, d, h, w = x_cl.shape[:4]
    if out is None:
        out = torch.zeros((n, d, h, w, cout), dtype=x_cl.dtype, device=DEV)
        out_pitch = cout
    tiles = lib().hdf_op_conv3d_stat_tiles(dtype, cin, d, h, w)
    part = torch.zeros((n * tiles, rup(cout, 32), 2), dtype=torch.float32, device=DEV)
    check(lib().hdf_op_conv3d_wr(dtype, ptr(x_cl), x_cl.shape[4], cin, n, d, h, w, ptr(w_packed), ptr(bias), ptr(scale),
                                 ptr(shift), relu, ptr(out), out_pitch, cout, ptr(part), accumulate, st()), "conv3d_wr")
    return out, part


@pytest.mark.parametrize("dtype", [BF16, F16])
@pytest.mark.parametrize("xf", [False])
@pytest.mark.parametrize("cin,cout,size,n", [(64, 32, (64, 56, 48), 2),      # K split over the waves + LDS exchange
                                            (64, 32, (56, 52, 50), 2),      # ragged in y and x, two samples
                                            (64, 64, (48, 52, 56), 1),      # two output blocks through blockIdx.y
                                            (64, 48, (50, 49, 51), 1)])     # channel count below the padded block
def test_conv3d_weights_in_registers(dtype, xf, cin, cout, size, n):
    """conv_wr_kernel (csrc/conv_wr.hip) through hdf_op_conv3d_wr: 16-bit storage, 128-byte rows at >= 48^3, no input
    transform -- the one instantiation per storage type that is built (round 5 dropped the five the plan never routed:
    64-byte rows and the in-place LDS transform; the `xf` branch below is kept for the day one returns) --, ragged extents
    (the checked border phase), several samples (the tile list, the deferred epilogue and the statistics rows cross
    samples), bias, InstanceNorm partial sums.  Reference: torch fp32 conv3d on the storage-rounded operands."""
    x, w, b = _mk((n, cin) + size, 21), _mk((cout, cin, 3, 3, 3), 22) * (cin * 27) ** -0.5, _mk((cout,), 23)
    scale, shift = _mk((n, cin), 24) * 0.5 + 1.0, _mk((n, cin), 25) * 0.3
    xa = rnd(x, dtype)
    if xf:
        xa = rnd(torch.relu(xa * scale[:, :, None, None, None] + shift[:, :, None, None, None]), dtype)
    ref = F.conv3d(xa, rnd(w, dtype), b, padding=1)
    wp = pack_w(w, dtype, cout, cin, rup(cout, 32), cin, cin * 27, 27, 0)
    sc, sh = (scale.to(DEV), shift.to(DEV)) if xf else (None, None)
    out, part = _conv3d_wr(dtype, to_cl(x, dtype), cin, wp, cout, bias=b.to(DEV), scale=sc, shift=sh, relu=1)
    torch.cuda.synchronize()
    got = from_cl(out)
    assert rel_err(got, ref) < TOL[dtype]
    # element-wise too: a misplaced tile or M-block would hide in a norm of 10^6 values
    assert float((got - ref).abs().max()) < 0.05 * float(ref.abs().max())
    tiles = part.shape[0] // n
    s = part.view(n, tiles, -1, 2).sum(1).cpu()[:, :cout]
    assert rel_err(s[..., 0], ref.sum((2, 3, 4))) < 5e-3 + TOL[dtype]
    assert rel_err(s[..., 1], (ref * ref).sum((2, 3, 4))) < 5e-3 + TOL[dtype]


@pytest.mark.parametrize("cin,cout,size,n", [(64, 32, (48, 48, 56), 2), (64, 64, (56, 48, 48), 1)])
def test_conv3d_weights_in_registers_accumulate_and_pitch(cin, cout, size, n):
    """out += conv(x) into a channel slice of a wider buffer, input a slice too: the launch is not `plain`, every tile runs
    the checked phase with the immediate epilogue."""
    dtype = BF16
    x, w = _mk((n, cin) + size, 31), _mk((cout, cin, 3, 3, 3), 32) * (cin * 27) ** -0.5
    base = _mk((n, cout) + size, 33)
    ref = rnd(base, dtype) + F.conv3d(rnd(x, dtype), rnd(w, dtype), None, padding=1)
    wp = pack_w(w, dtype, cout, cin, rup(cout, 32), cin, cin * 27, 27, 0)
    xcl = to_cl(x, dtype)
    wide_in = torch.zeros((n,) + size + (2 * cin,), dtype=xcl.dtype, device=DEV)
    wide_in[..., cin:] = xcl
    wide_out = torch.full((n,) + size + (3 * cout,), 7.0, dtype=xcl.dtype, device=DEV)
    wide_out[..., cout:2 * cout] = to_cl(base, dtype)
    vin, vout = wide_in.view(-1)[cin:], wide_out.view(-1)[cout:]
    check(lib().hdf_op_conv3d_wr(dtype, ptr(vin), 2 * cin, cin, n, *size, ptr(wp), None, None, None, 0, ptr(vout),
                                 3 * cout, cout, None, 1, st()), "conv3d_wr")
    torch.cuda.synchronize()
    assert rel_err(from_cl(wide_out[..., cout:2 * cout]), ref) < TOL[dtype] * 1.5
    assert float((wide_out[..., :cout].float() - 7.0).abs().max()) == 0.0      # the neighbouring slices are untouched
    assert float((wide_out[..., 2 * cout:].float() - 7.0).abs().max()) == 0.0


@pytest.mark.parametrize("dtype", [F32, BF16, F16])
@pytest.mark.parametrize("cin,cout,size,n", [(32, 32, (8, 8, 16), 2), (32, 64, (48, 48, 52), 3), (64, 32, (52, 48, 48), 1),
                                            (16, 48, (48, 50, 48), 1)])
def test_conv3d_accumulate_into_existing_gradient(dtype, cin, cout, size, n):
    """out += conv(x): the dgrad of the UpConv chain adds into a skip gradient (plan.hip conv_backward, accumulate=1).
    >= 48^3 runs the persistent kernel's non-deferred epilogue (batch 3: the tile list crosses samples)."""
    x, w = _mk((n, cin) + size, 11), _mk((cout, cin, 3, 3, 3), 12) * (cin * 27) ** -0.5
    base = _mk((n, cout) + size, 13)
    ref = rnd(base, dtype) + F.conv3d(rnd(x, dtype), rnd(w, dtype), None, padding=1)
    wp = pack_w(w, dtype, cout, cin, rup(cout, 32), cin, cin * 27, 27, 0)
    out = to_cl(base, dtype).contiguous()
    conv3d(dtype, 0, to_cl(x, dtype), cin, wp, cout, out=out, out_pitch=cout, accumulate=1)
    torch.cuda.synchronize()
    assert rel_err(from_cl(out), ref) < TOL[dtype] * 1.5


@pytest.mark.parametrize("dtype", [F32, BF16, F16])
@pytest.mark.parametrize("n,cin,cout,size", [(2, 32, 32, (8, 16, 8)), (2, 32, 32, (48, 48, 56)), (1, 64, 32, (50, 48, 48))])
def test_conv3d_input_transform_and_pitch(dtype, n, cin, cout, size):
    """producer's InstanceNorm+ReLU applied on load; input/output are channel slices of wider buffers."""
    x, w = _mk((n, cin) + size, 4), _mk((cout, cin, 3, 3, 3), 5) * (cin * 27) ** -0.5
    scale, shift = _mk((n, cin), 6) * 0.5 + 1.0, _mk((n, cin), 7) * 0.3
    xa = torch.relu(rnd(x, dtype) * scale[:, :, None, None, None] + shift[:, :, None, None, None])
    ref = F.conv3d(rnd(xa, dtype), rnd(w, dtype), None, padding=1)
    xcl = to_cl(x, dtype)
    wide_in = torch.zeros((n,) + size + (2 * cin,), dtype=xcl.dtype, device=DEV)
    wide_in[..., cin:] = xcl
    wide_out = torch.full((n,) + size + (3 * cout,), 7.0, dtype=xcl.dtype, device=DEV)
    wp = pack_w(w, dtype, cout, cin, rup(cout, 32), cin, cin * 27, 27, 0)
    vin = wide_in.view(-1)[cin:]
    vout = wide_out.view(-1)[cout:]
    sc, sh = scale.to(DEV), shift.to(DEV)
    check(lib().hdf_op_conv3d(dtype, 0, ptr(vin), 2 * cin, cin, n, *size, ptr(wp), None, ptr(sc), ptr(sh), 1,
                              ptr(vout), 3 * cout, cout, None, 0, st()), "conv")
    torch.cuda.synchronize()
    got = from_cl(wide_out[..., cout:2 * cout])
    assert rel_err(got, ref) < TOL[dtype] * 1.5
    assert float((wide_out[..., :cout].float() - 7).abs().max()) == 0
    assert float((wide_out[..., 2 * cout:].float() - 7).abs().max()) == 0


@pytest.mark.parametrize("dtype", [F32, BF16, F16])
@pytest.mark.parametrize("cin,cout,size", [(32, 16, (4, 4, 4)), (64, 32, (8, 8, 8)), (128, 64, (6, 5, 7))])
def test_conv_transpose3d(dtype, cin, cout, size):
    n = 2
    x, w, b = _mk((n, cin) + size, 8), _mk((cin, cout, 3, 3, 3), 9) * (cin * 27 / 8) ** -0.5, _mk((cout,), 10)
    ref = F.conv_transpose3d(rnd(x, dtype), rnd(w, dtype), b, stride=2, padding=1, output_padding=1)
    wp = pack_w(w, dtype, cout, cin, rup(cout, 32), cin, 27, cout * 27, 0)
    out, _ = conv3d(dtype, 2, to_cl(x, dtype), cin, wp, cout, bias=b.to(DEV))
    torch.cuda.synchronize()
    assert rel_err(from_cl(out), ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("cin,cout,split,size", [(32, 64, 32, (48, 48, 48)),   # conv_ws2, two output blocks per workgroup
                                                  (64, 128, 64, (48, 56, 48)),  # conv_ws2, 128-byte rows
                                                  (64, 64, 32, (16, 24, 16))])  # conv_igemm
def test_conv3d_split_output_and_column_sums(dtype, cin, cout, split, size):
    """The dgrad form of a decoder concat: one conv, output channels [0, split) and [split, cout) in two dense buffers,
    plus the per-channel sums of the first half from the statistics epilogue (the ConvTranspose3d bias gradient)."""
    n = 2
    x, w = _mk((n, cin) + size, 61), _mk((cout, cin, 3, 3, 3), 62) * (cin * 27) ** -0.5
    ref = F.conv3d(rnd(x, dtype), rnd(w, dtype), None, padding=1)
    wp = pack_w(w, dtype, cout, cin, rup(cout, 32), cin, cin * 27, 27, 0)
    x_cl = to_cl(x, dtype)
    o1 = torch.full((n,) + size + (split,), 7.0, dtype=x_cl.dtype, device=DEV)
    o2 = torch.full((n,) + size + (cout - split,), 7.0, dtype=x_cl.dtype, device=DEV)
    assert split == cout - split                      # one pitch for both buffers
    tiles = lib().hdf_op_conv3d_stat_tiles(dtype, cin, *size)
    part = torch.zeros((n * tiles, rup(cout, 32), 2), dtype=torch.float32, device=DEV)
    colsum = torch.zeros(split, dtype=torch.float32, device=DEV)
    check(lib().hdf_op_conv3d_split(dtype, ptr(x_cl), cin, cin, n, *size, ptr(wp), ptr(o1), ptr(o2), split, cout, split,
                                    ptr(part), ptr(colsum), split, st()), "conv3d_split")
    torch.cuda.synchronize()
    assert rel_err(from_cl(o1), ref[:, :split]) < TOL[dtype]
    assert rel_err(from_cl(o2), ref[:, split:]) < TOL[dtype]
    want = ref[:, :split].double().sum(dim=(0, 2, 3, 4))
    got = colsum.cpu().double()
    assert float((got - want).abs().max() / (ref[:, :split].double().abs().sum(dim=(0, 2, 3, 4)).max())) < TOL[dtype]


@pytest.mark.parametrize("dtype", [BF16, F16])
@pytest.mark.parametrize("cout,size,n", [(32, (8, 8, 16), 2), (16, (4, 12, 8), 1), (32, (20, 4, 8), 3)])
@pytest.mark.parametrize("xf", [0, 1])
def test_conv_transpose3d_whole_tiles(dtype, cout, size, n, xf):
    """64 input channels, extents multiples of (4, 4, 8): the persistent weights-in-registers transposed conv (several
    tiles per workgroup and samples, high-face zero padding, a partial output-channel block, the in-place
    InstanceNorm+ReLU transform of its input), written into the up-convolution half of a wider row."""
    cin = 64
    x, w, b = _mk((n, cin) + size, 51), _mk((cin, cout, 3, 3, 3), 52) * (cin * 27 / 8) ** -0.5, _mk((cout,), 53)
    sc = (torch.rand(n, cin, generator=torch.Generator().manual_seed(54)) + 0.5) if xf else None
    sh = (torch.randn(n, cin, generator=torch.Generator().manual_seed(55)) * 0.3) if xf else None
    xin = rnd(x, dtype)
    if xf:
        xin = rnd(torch.relu(xin * sc[:, :, None, None, None] + sh[:, :, None, None, None]), dtype)
    ref = F.conv_transpose3d(xin, rnd(w, dtype), b, stride=2, padding=1, output_padding=1)
    wp = pack_w(w, dtype, cout, cin, rup(cout, 32), cin, 27, cout * 27, 0)
    osz = tuple(2 * v for v in size)
    wide = torch.full((n,) + osz + (2 * cout,), 7.0, dtype=to_cl(x, dtype).dtype, device=DEV)
    conv3d(dtype, 2, to_cl(x, dtype), cin, wp, cout, bias=b.to(DEV), scale=sc.to(DEV) if xf else None,
           shift=sh.to(DEV) if xf else None, relu=xf, out=wide, out_pitch=2 * cout)
    torch.cuda.synchronize()
    assert rel_err(from_cl(wide[..., :cout]), ref) < TOL[dtype]
    assert float((wide[..., cout:].float() - 7).abs().max()) == 0


@pytest.mark.parametrize("dtype", [BF16, F16])
def test_transposed_conv_kernels_with_an_odd_channel_count_take_the_generic_path(dtype):
    """The two persistent top-level transposed-conv kernels store channel PAIRS as one dword (st_rows2): an odd channel
    count or pitch, legal at the operator ABI, must fall through to the element-store kernels, not write channel Cout
    (= channel 0 of the next voxel).  Forward (64 -> 31 channels, pitch 31) and its data gradient (31 -> 63, pitch 63)."""
    n, size = 1, (4, 4, 8)
    for mode, cin, cout in ((2, 64, 31), (1, 32, 63)):
        if mode == 2:
            x, w = _mk((n, cin) + size, 61), _mk((cin, cout, 3, 3, 3), 62) * (cin * 27 / 8) ** -0.5
            ref = F.conv_transpose3d(rnd(x, dtype), rnd(w, dtype), None, stride=2, padding=1, output_padding=1)
            wp = pack_w(w, dtype, cout, cin, rup(cout, 32), cin, 27, cout * 27, 0)
            osz = tuple(2 * v for v in size)
        else:
            big = tuple(2 * v for v in size)
            x, w = _mk((n, cin) + big, 63), _mk((cout, cin, 3, 3, 3), 64) * (cin * 27) ** -0.5
            ref = F.conv3d(rnd(x, dtype), rnd(w, dtype), None, stride=2, padding=1)
            wp = pack_w(w, dtype, cout, cin, rup(cout, 32), cin, cin * 27, 27, 0)
            osz = size
        out = torch.full((n,) + osz + (cout,), 7.0, dtype=to_cl(x, dtype).dtype, device=DEV)
        guard = torch.full((64,), 7.0, dtype=out.dtype, device=DEV)          # allocated right behind it
        conv3d(dtype, mode, to_cl(x, dtype), cin, wp, cout, out=out, out_pitch=cout)
        torch.cuda.synchronize()
        assert rel_err(from_cl(out), ref) < TOL[dtype], (mode, cin, cout)
        assert float((guard.float() - 7).abs().max()) == 0


@pytest.mark.parametrize("dtype", [F32, BF16, F16])
@pytest.mark.parametrize("cin,cout,size", [(16, 32, (8, 8, 8)), (32, 64, (16, 8, 12)), (64, 128, (10, 12, 6))])
def test_conv3d_stride2(dtype, cin, cout, size):
    """stride-2 gather conv == dgrad of ConvTranspose3d(k3,s2,p1,op1)"""
    n = 1
    x, w = _mk((n, cin) + size, 11), _mk((cout, cin, 3, 3, 3), 12) * (cin * 27) ** -0.5
    ref = F.conv3d(rnd(x, dtype), rnd(w, dtype), None, stride=2, padding=1)
    wp = pack_w(w, dtype, cout, cin, rup(cout, 32), cin, cin * 27, 27, 0)
    out, _ = conv3d(dtype, 1, to_cl(x, dtype), cin, wp, cout)
    torch.cuda.synchronize()
    assert rel_err(from_cl(out), ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", [BF16, F16])
@pytest.mark.parametrize("cout,size,n", [(64, (16, 16, 24), 2), (48, (8, 16, 8), 1), (64, (40, 8, 8), 3)])
def test_conv3d_stride2_whole_tiles(dtype, cout, size, n):
    """32 -> 64 channels, output extents multiples of 4: the persistent weights-in-registers gather kernel (several
    tiles per workgroup and samples, low-face zero padding, a partial output-channel block, bias)."""
    cin = 32
    x, w = _mk((n, cin) + size, 41), _mk((cout, cin, 3, 3, 3), 42) * (cin * 27) ** -0.5
    b = _mk((cout,), 43) * 0.1
    ref = F.conv3d(rnd(x, dtype), rnd(w, dtype), b, stride=2, padding=1)
    wp = pack_w(w, dtype, cout, cin, rup(cout, 32), cin, cin * 27, 27, 0)
    out, _ = conv3d(dtype, 1, to_cl(x, dtype), cin, wp, cout, bias=b.to(DEV))
    torch.cuda.synchronize()
    assert rel_err(from_cl(out), ref) < TOL[dtype]


def _wgrad(dtype, stride, s_cl, sc, l_cl, lc, dims, sc_store, lc_store, s_scale=None, s_shift=None, s_relu=0):
    n = s_cl.shape[0]
    wsb = lib().hdf_op_wgrad_workspace_bytes(stride, n, *dims, sc, lc)
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    dw = torch.zeros((sc_store, lc_store, 27), dtype=torch.float32, device=DEV)
    check(lib().hdf_op_conv3d_wgrad(dtype, stride, ptr(s_cl), s_cl.shape[-1], sc, ptr(l_cl), l_cl.shape[-1], lc, n,
                                    *dims, ptr(s_scale), ptr(s_shift), s_relu, None, None, 0, ptr(dw), sc_store, lc_store,
                                    0, ptr(ws), wsb, st()), "wgrad")
    torch.cuda.synchronize()
    return dw.cpu()


@pytest.mark.parametrize("dtype", [F32, BF16, F16])
@pytest.mark.parametrize("cin,cout,size,n", [(16, 16, (8, 8, 8), 1), (32, 32, (12, 16, 24), 2),
                                            (64, 48, (9, 7, 10), 2),
                                            # 256+ tiles: the interior / border passes and the XCD-interleaved split
                                            (32, 32, (32, 40, 48), 2), (64, 32, (21, 48, 50), 3)])
def test_conv3d_wgrad(dtype, cin, cout, size, n):
    x, dy = _mk((n, cin) + size, 13), _mk((n, cout) + size, 14)
    w = torch.zeros(cout, cin, 3, 3, 3, requires_grad=True)
    F.conv3d(rnd(x, dtype), w, None, padding=1).backward(rnd(dy, dtype))
    got = _wgrad(dtype, 1, to_cl(dy, dtype), cout, to_cl(x, dtype), cin, size, cout, cin).view(cout, cin, 3, 3, 3)
    assert rel_err(got, w.grad) < TOL[dtype]


@pytest.mark.parametrize("dtype", [F32, BF16, F16])
def test_conv3d_wgrad_first_layer_padded_input(dtype):
    """Cin=4 input padded to 16 channels; only the 4 real input channels are stored."""
    n, cin, cout, size = 1, 4, 32, (8, 8, 16)
    x, dy = _mk((n, cin) + size, 15), _mk((n, cout) + size, 16)
    w = torch.zeros(cout, cin, 3, 3, 3, requires_grad=True)
    F.conv3d(rnd(x, dtype), w, None, padding=1).backward(rnd(dy, dtype))
    got = _wgrad(dtype, 1, to_cl(dy, dtype), cout, to_cl(x, dtype, cp=16), 16, size, cout, cin)
    assert rel_err(got.view(cout, cin, 3, 3, 3), w.grad) < TOL[dtype]


@pytest.mark.parametrize("dtype", [F32, BF16, F16])
@pytest.mark.parametrize("cin,cout,size", [(32, 16, (4, 4, 4)), (64, 32, (6, 5, 7))])
def test_conv_transpose3d_wgrad(dtype, cin, cout, size):
    n = 2
    osz = tuple(2 * s for s in size)
    x, dy = _mk((n, cin) + size, 17), _mk((n, cout) + osz, 18)
    w = torch.zeros(cin, cout, 3, 3, 3, requires_grad=True)
    F.conv_transpose3d(rnd(x, dtype), w, None, stride=2, padding=1, output_padding=1).backward(rnd(dy, dtype))
    got = _wgrad(dtype, 2, to_cl(x, dtype), cin, to_cl(dy, dtype), cout, size, cin, cout).view(cin, cout, 3, 3, 3)
    assert rel_err(got, w.grad) < TOL[dtype]


@pytest.mark.parametrize("dtype", [BF16, F16])
@pytest.mark.parametrize("cin,cout,size,n", [(64, 32, (8, 12, 16), 2), (96, 48, (8, 8, 8), 1), (128, 64, (16, 16, 16), 3),
                                            (32, 32, (20, 8, 12), 2), (64, 16, (4, 4, 8), 1)])
@pytest.mark.parametrize("xf", [0, 1])
def test_conv_transpose3d_wgrad_whole_tiles(dtype, cin, cout, size, n, xf):
    """Extents that are multiples of 4: the double-buffered stride-2 kernel (one and two input-channel blocks per
    workgroup, partial channel blocks, several tiles per workgroup and samples), without / with the InstanceNorm+ReLU
    transform of the transposed conv's input."""
    osz = tuple(2 * s for s in size)
    x, dy = _mk((n, cin) + size, 27), _mk((n, cout) + osz, 28)
    sc = (torch.rand(n, cin, generator=torch.Generator().manual_seed(29)) + 0.5) if xf else None
    sh = (torch.randn(n, cin, generator=torch.Generator().manual_seed(30)) * 0.3) if xf else None
    xin = rnd(x, dtype)
    if xf:
        xin = rnd(torch.relu(xin * sc[:, :, None, None, None] + sh[:, :, None, None, None]), dtype)
    w = torch.zeros(cin, cout, 3, 3, 3, requires_grad=True)
    F.conv_transpose3d(xin, w, None, stride=2, padding=1, output_padding=1).backward(rnd(dy, dtype))
    got = _wgrad(dtype, 2, to_cl(x, dtype), cin, to_cl(dy, dtype), cout, size, cin, cout,
                 sc.to(DEV) if xf else None, sh.to(DEV) if xf else None, xf).view(cin, cout, 3, 3, 3)
    assert rel_err(got, w.grad) < TOL[dtype]


@pytest.mark.parametrize("dtype", [F32, BF16, F16])
def test_pool_upsample(dtype):
    n, c, size = 2, 32, (8, 12, 16)
    x = _mk((n, c) + size, 19)
    x[:, :, :2] = 0.0                                  # exact ties: the FIRST max must win (torch rule)
    xr = rnd(x, dtype).requires_grad_(True)
    x_cl = to_cl(x, dtype)
    # --- maxpool fwd/bwd
    po = torch.empty((n,) + tuple(s // 2 for s in size) + (c,), dtype=x_cl.dtype, device=DEV)
    idx = torch.empty(po.shape, dtype=torch.uint8, device=DEV)
    check(lib().hdf_op_maxpool_fwd(dtype, ptr(x_cl), c, ptr(po), c, ptr(idx), n, c, *po.shape[1:4], st()), "pool")
    ref = F.max_pool3d(xr, 2)
    g = _mk(tuple(ref.shape), 20)
    ref.backward(rnd(g, dtype))
    din = torch.zeros_like(x_cl)
    gcl = to_cl(g, dtype)
    check(lib().hdf_op_maxpool_bwd(dtype, ptr(gcl), c, ptr(idx), ptr(din), c, n, c, *po.shape[1:4], 0, st()), "poolb")
    torch.cuda.synchronize()
    assert rel_err(from_cl(po), ref.detach()) == 0
    assert rel_err(from_cl(din), xr.grad) < 1e-6
    # --- trilinear x2 of relu(x*scale+shift), fwd/bwd
    scale, shift = _mk((n, c), 21) * 0.5 + 1.0, _mk((n, c), 22) * 0.3
    xa = torch.relu(rnd(x, dtype) * scale[:, :, None, None, None] + shift[:, :, None, None, None]).detach().requires_grad_(True)
    ref = F.interpolate(xa, scale_factor=2, mode="trilinear", align_corners=False)
    up = torch.empty((n,) + tuple(2 * s for s in size) + (c,), dtype=x_cl.dtype, device=DEV)
    sc, sh = scale.to(DEV), shift.to(DEV)
    check(lib().hdf_op_upsample_fwd(dtype, ptr(x_cl), c, ptr(sc), ptr(sh), ptr(up), c, n, c, *size, st()), "up")
    g2 = _mk(tuple(ref.shape), 23)
    ref.backward(rnd(g2, dtype))
    dlo = torch.empty_like(x_cl)
    g2cl = to_cl(g2, dtype)
    check(lib().hdf_op_upsample_bwd(dtype, ptr(g2cl), c, ptr(dlo), c, n, c, *size, st()), "upb")
    torch.cuda.synchronize()
    assert rel_err(from_cl(up), ref.detach()) < TOL[dtype]
    assert rel_err(from_cl(dlo), xa.grad) < TOL[dtype]


@pytest.mark.parametrize("dtype", [F32, BF16, F16])
@pytest.mark.parametrize("c", [16, 32, 48])
def test_maxpool_backward_with_instance_norm_first_pass(dtype, c):
    """hdf_op_maxpool_bwd_in: din += scatter(dout) exactly as hdf_op_maxpool_bwd(accumulate=1), and the partial rows sum
    to (sum g, sum g * xhat) over the voxels where relu(y * scale + shift) is positive, g = the stored din."""
    n, size = 2, (8, 12, 16)
    ps = tuple(v // 2 for v in size)
    x = _mk((n, c) + size, 41)
    x_cl = to_cl(x, dtype)
    po = torch.empty((n,) + ps + (c,), dtype=x_cl.dtype, device=DEV)
    idx = torch.empty(po.shape, dtype=torch.uint8, device=DEV)
    check(lib().hdf_op_maxpool_fwd(dtype, ptr(x_cl), c, ptr(po), c, ptr(idx), n, c, *ps, st()), "pool")
    gcl = to_cl(_mk((n, c) + ps, 42), dtype)
    d0 = to_cl(_mk((n, c) + size, 43), dtype)
    y = _mk((n, c) + size, 44)
    y_cl = to_cl(y, dtype)
    yr = from_cl(y_cl)
    mean, rstd = yr.mean((2, 3, 4)), (yr.var((2, 3, 4), unbiased=False) + 1e-5).rsqrt()
    gamma, beta = _mk((c,), 45) * 0.3 + 1.0, _mk((c,), 46) * 0.2
    gamma[0] = -gamma[0]
    scale, shift = (gamma[None] * rstd).contiguous(), (beta[None] - mean * gamma[None] * rstd).contiguous()
    dev = [t.to(DEV).contiguous() for t in (scale, shift, mean, rstd)]
    ref = d0.clone()
    check(lib().hdf_op_maxpool_bwd(dtype, ptr(gcl), c, ptr(idx), ptr(ref), c, n, c, *ps, 1, st()), "poolb")
    got = d0.clone()
    rows = lib().hdf_op_maxpool_bwd_in_rows(c, *ps)
    part = torch.full((n, rows, c, 2), float("nan"), device=DEV)
    check(lib().hdf_op_maxpool_bwd_in(dtype, ptr(gcl), c, ptr(idx), ptr(got), c, ptr(y_cl), c, *[ptr(t) for t in dev],
                                      ptr(part), n, c, *ps, st()), "poolb_in")
    torch.cuda.synchronize()
    assert bool((got == ref).all())
    g = from_cl(got).double()
    act = yr.double() * scale.double()[:, :, None, None, None] + shift.double()[:, :, None, None, None]
    gm = torch.where(act > 0, g, torch.zeros_like(g))
    xh = (yr.double() - mean.double()[:, :, None, None, None]) * rstd.double()[:, :, None, None, None]
    s = part.double().sum(1).cpu()
    assert rel_err(s[..., 0], gm.sum((2, 3, 4))) < 1e-4 and rel_err(s[..., 1], (gm * xh).sum((2, 3, 4))) < 1e-4


@pytest.mark.parametrize("dtype", [BF16, F16])
@pytest.mark.parametrize("cin,cout,size,n", [(4, 32, (8, 16, 24), 2), (4, 32, (5, 9, 11), 1), (1, 16, (4, 8, 8), 2),
                                             (3, 48, (6, 10, 12), 1), (2, 64, (8, 8, 16), 1), (4, 32, (32, 32, 32), 1)])
@pytest.mark.parametrize("with_bias", [False, True])
def test_conv3d_first_layer_tap_packed(dtype, cin, cout, size, n, with_bias):
    """hdf_op_conv3d_first (csrc/conv_first.hip): the encoder's first Conv3d(in_channels <= 4, n_filters, 3, padding=1)
    (HDenseFormer.py:152-158,190) with K = (tap, channel), against torch on the storage-rounded operands; the input is the
    plan's 16-channel row of which the first `cin` channels are real (the rest poisoned here: they must not be read into
    the result), ragged extents, 1..4 channels, 16..64 filters, and the InstanceNorm partial sums of the fp32 results."""
    x = _mk((n, cin) + size, 51)
    w = _mk((cout, cin, 3, 3, 3), 52) * 0.2
    b = _mk((cout,), 53) * 0.1 if with_bias else None
    ref = F.conv3d(rnd(x, dtype), rnd(w, dtype), b, padding=1)
    pitch = 16
    xin = torch.full((n,) + size + (pitch,), 7.0, dtype=TDT[dtype], device=DEV)
    xin[..., :cin] = x.to(DEV).permute(0, 2, 3, 4, 1).to(TDT[dtype])
    if cin < 4:
        xin[..., cin:4] = 0       # the plan's padded channels are zeros (the kernel reads 4 channels and masks by Cin)
    coutp = (cout + 31) // 32 * 32
    out = torch.full((n,) + size + (coutp,), float("nan"), dtype=TDT[dtype], device=DEV)
    part = torch.full((n, 512, coutp, 2), float("nan"), device=DEV)
    wd = w.to(DEV).contiguous()
    bd = b.to(DEV).contiguous() if with_bias else None
    check(lib().hdf_op_conv3d_first(dtype, ptr(xin), pitch, cin, n, *size, ptr(wd), ptr(bd), ptr(out), coutp, cout, ptr(part),
                                    st()), "conv3d_first")
    torch.cuda.synchronize()
    got = out[..., :cout].permute(0, 4, 1, 2, 3).float().cpu()
    assert rel_err(got, ref) < TOL[dtype]
    s = part.double().sum(1).cpu()
    assert rel_err(s[:, :cout, 0], ref.double().sum((2, 3, 4))) < 1e-3
    assert rel_err(s[:, :cout, 1], (ref.double() ** 2).sum((2, 3, 4))) < 1e-3
    assert bool((s[:, cout:] == 0).all())
    if coutp > cout:
        assert bool(torch.isnan(out[..., cout:].float()).all())     # channels beyond Cout are not written


@pytest.mark.parametrize("dtype", [BF16, F16])
@pytest.mark.parametrize("cin,cout,size,n", [(4, 32, (8, 16, 24), 2), (4, 32, (5, 9, 11), 1), (1, 16, (4, 8, 8), 2),
                                             (3, 48, (6, 10, 12), 1), (2, 64, (8, 8, 16), 1), (4, 32, (32, 32, 32), 2)])
@pytest.mark.parametrize("accumulate", [0, 1])
def test_conv3d_first_layer_weight_gradient(dtype, cin, cout, size, n, accumulate):
    """hdf_op_conv3d_first_wgrad (csrc/conv_first.hip): the weight gradient of the encoder's first Conv3d against autograd
    on the storage-rounded operands (HDenseFormer.py:152-158,190), dy a channel slice of a wider buffer, ragged extents,
    1..4 input channels, 16..64 filters, overwrite and accumulate."""
    x = rnd(_mk((n, cin) + size, 61), dtype)
    dy = rnd(_mk((n, cout) + size, 62), dtype)
    w = torch.zeros(cout, cin, 3, 3, 3, requires_grad=True)
    F.conv3d(x, w, None, padding=1).backward(dy)
    xin = torch.full((n,) + size + (16,), 7.0, dtype=TDT[dtype], device=DEV)
    xin[..., :4] = 0
    xin[..., :cin] = x.to(DEV).permute(0, 2, 3, 4, 1).to(TDT[dtype])
    pitch = cout + 16
    dyb = torch.full((n,) + size + (pitch,), 5.0, dtype=TDT[dtype], device=DEV)
    dyb[..., 8:8 + cout] = dy.to(DEV).permute(0, 2, 3, 4, 1).to(TDT[dtype])
    dyv = dyb.view(-1)[8:]
    prev = _mk((cout, cin, 3, 3, 3), 63)
    dw = prev.to(DEV).contiguous() if accumulate else torch.full((cout, cin, 3, 3, 3), float("nan"), device=DEV)
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=DEV)
    check(lib().hdf_op_conv3d_first_wgrad(dtype, ptr(dyv), pitch, cout, ptr(xin), 16, cin, n, *size, ptr(dw), accumulate,
                                          ptr(ws), ws.numel(), st()), "conv3d_first_wgrad")
    torch.cuda.synchronize()
    ref = w.grad + (prev if accumulate else 0)
    assert rel_err(dw.cpu(), ref) < 1e-4


@pytest.mark.parametrize("dtype", [BF16, F16])
@pytest.mark.parametrize("cin,cout,size,n", [(4, 32, (8, 16, 24), 2), (4, 32, (5, 9, 11), 1), (3, 48, (6, 10, 12), 1),
                                             (2, 64, (8, 8, 16), 2)])
def test_conv3d_first_layer_weight_gradient_with_the_norm_backward_inside(dtype, cin, cout, size, n):
    """hdf_op_conv3d_first_wgrad_in: the first layer's weight gradient from d(activation) and y, with the apply pass of the
    InstanceNorm(+ReLU) backward (unet_ops.hip in_bwd_apply4_kernel) evaluated on the staged rows -- against the two-step
    path: dy by that formula in torch (rounded to the storage type), then hdf_op_conv3d_first_wgrad on it."""
    x = rnd(_mk((n, cin) + size, 71), dtype)
    da, y = rnd(_mk((n, cout) + size, 72), dtype), rnd(_mk((n, cout) + size, 73), dtype)
    vec = [(_mk((n, cout), 74 + i) * 0.3 + (1.0 if i in (0, 3, 4) else 0.0)).contiguous() for i in range(7)]
    sc, sh, mu, rs, k1, ka, kb = vec
    bc = lambda v: v[:, :, None, None, None]
    gg = torch.where(y * bc(sc) + bc(sh) > 0, da, torch.zeros_like(da))
    dy = rnd(bc(k1) * (gg - bc(ka) - (y - bc(mu)) * bc(rs) * bc(kb)), dtype)
    xin = torch.zeros((n,) + size + (16,), dtype=TDT[dtype], device=DEV)
    xin[..., :cin] = x.to(DEV).permute(0, 2, 3, 4, 1).to(TDT[dtype])
    cl = lambda t: t.to(DEV).permute(0, 2, 3, 4, 1).to(TDT[dtype]).contiguous()
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=DEV)
    ref = torch.full((cout, cin, 3, 3, 3), float("nan"), device=DEV)
    dy_cl = cl(dy)
    check(lib().hdf_op_conv3d_first_wgrad(dtype, ptr(dy_cl), cout, cout, ptr(xin), 16, cin, n, *size, ptr(ref), 0, ptr(ws),
                                          ws.numel(), st()), "first_wgrad")
    got = torch.full((cout, cin, 3, 3, 3), float("nan"), device=DEV)
    da_cl, y_cl = cl(da), cl(y)
    dev = [v.to(DEV) for v in vec]
    check(lib().hdf_op_conv3d_first_wgrad_in(dtype, ptr(da_cl), cout, cout, ptr(y_cl), cout, *[ptr(v) for v in dev], ptr(xin),
                                             16, cin, n, *size, ptr(got), 0, ptr(ws), ws.numel(), st()), "first_wgrad_in")
    torch.cuda.synchronize()
    assert rel_err(got.cpu(), ref.cpu()) < 2e-3     # (a value at a rounding tie of the storage type may fall the other way)


@pytest.mark.parametrize("dtype", [BF16, F16])
@pytest.mark.parametrize("size,n", [((48, 48, 48), 2), ((52, 48, 56), 1)])
def test_conv3d_data_gradient_with_the_next_norm_backward_statistics(dtype, size, n):
    """hdf_op_conv3d_bwd_stats: the 32 -> 32 stride-1 conv as a data gradient whose epilogue writes the first pass of the next
    InstanceNorm(+ReLU) backward (csrc/conv_igemm.hip conv_ws2_kernel<..., BS>): the output must equal the plain conv's bit for
    bit, and the partial rows summed over the workgroups must be (sum g, sum g * xhat) of the STORED output."""
    cin = cout = 32
    x = _mk((n, cin) + size, 81)
    w = _mk((cout, cin, 3, 3, 3), 82) * 0.1
    y = rnd(_mk((n, cout) + size, 83), dtype)
    sc, sh = _mk((n, cout), 84) * 0.5 + 1.0, _mk((n, cout), 85) * 0.3
    mu, rs = _mk((n, cout), 86) * 0.2, _mk((n, cout), 87).abs() + 0.5
    wp = pack_w(w, dtype, cout, cin, 32, cin, cin * 27, 27, 0)
    x_cl, y_cl = to_cl(x, dtype), to_cl(y, dtype)
    ref_out, _ = conv3d(dtype, 0, x_cl, cin, wp, cout, stats=True)
    out = torch.full_like(ref_out, float("nan"))
    part = torch.full((n, 512, cout, 2), float("nan"), device=DEV)
    dev = [v.to(DEV).contiguous() for v in (sc, sh, mu, rs)]
    check(lib().hdf_op_conv3d_bwd_stats(dtype, ptr(x_cl), cin, cin, n, *size, ptr(wp), ptr(out), cout, cout, ptr(y_cl), cout,
                                        *[ptr(v) for v in dev], ptr(part), st()), "conv3d_bwd_stats")
    torch.cuda.synchronize()
    assert torch.equal(out, ref_out)
    g = from_cl(out).double()
    bc = lambda v: v.double()[:, :, None, None, None]
    yd = y.double()
    gm = torch.where(yd * bc(sc) + bc(sh) > 0, g, torch.zeros_like(g))
    s = part.double().sum(1).cpu()
    assert rel_err(s[..., 0], gm.sum((2, 3, 4))) < 1e-4
    assert rel_err(s[..., 1], (gm * (yd - bc(mu)) * bc(rs)).sum((2, 3, 4))) < 1e-4


@pytest.mark.parametrize("dtype", [F32, BF16, F16])
@pytest.mark.parametrize("c", [16, 32, 48, 128])
def test_encoder_tail_vs_torch(dtype, c):
    """hdf_op_enc_tail: ds = relu(y * scale + shift) + skip and MaxPool3d(2) of the stored ds with torch's tie rule, in one
    pass (HDenseFormer.py:238-243), incl. a channel count that is not a power of two, exact ties (a block of zeros after the
    ReLU: the FIRST maximum in scan order must win) and a negative gamma."""
    n, size = 2, (8, 12, 16)
    y = _mk((n, c) + size, 31)
    skip = _mk((n, c) + size, 32)
    skip[:, :, :2] = 0.0
    y[:, :, :2] = -1.0                                  # relu -> exact zeros: ties, the FIRST maximum must win
    scale, shift = _mk((n, c), 33) * 0.5 + 1.0, _mk((n, c), 34) * 0.3
    scale[:, 0] = -scale[:, 0]                          # a negative gamma
    shift[:, 1] = -abs(shift[:, 1])                     # channel 1 of the zeroed planes: relu(-1 * s + t) == 0 exactly
    scale[:, 1] = abs(scale[:, 1])
    act = torch.relu(rnd(y, dtype) * scale[:, :, None, None, None] + shift[:, :, None, None, None]) + rnd(skip, dtype)
    ds_ref = rnd(act, dtype)
    y_cl, sk_cl = to_cl(y, dtype), to_cl(skip, dtype)
    ds = torch.empty_like(y_cl)
    po = torch.empty((n,) + tuple(v // 2 for v in size) + (c,), dtype=y_cl.dtype, device=DEV)
    idx = torch.empty(po.shape, dtype=torch.uint8, device=DEV)
    sc, sh = scale.to(DEV).contiguous(), shift.to(DEV).contiguous()
    check(lib().hdf_op_enc_tail(dtype, ptr(y_cl), c, ptr(sc), ptr(sh), ptr(sk_cl), c, ptr(ds), c, ptr(po), c, ptr(idx), n, c,
                                *po.shape[1:4], st()), "enc_tail")
    torch.cuda.synchronize()
    got_ds, got_po = from_cl(ds), from_cl(po)
    assert rel_err(got_ds, ds_ref) < TOL[dtype]
    # pooling is exact on the stored values: compare against a pool of OUR stored ds
    po2, idx2 = F.max_pool3d(got_ds, 2, return_indices=True)
    assert bool((got_po == po2).all())
    # arg-max byte -> flat index of the full-resolution volume, as torch reports it
    k = idx.permute(0, 4, 1, 2, 3).cpu().long()
    od, oh, ow = torch.meshgrid(*[torch.arange(v // 2) for v in size], indexing="ij")
    flat = ((2 * od + (k >> 2)) * size[1] + 2 * oh + ((k >> 1) & 1)) * size[2] + 2 * ow + (k & 1)
    assert bool((flat == idx2).all())
    assert int((got_ds[:, 1, :2] == 0).sum()) == got_ds[:, 1, :2].numel()      # the tie case is really in the data


@pytest.mark.parametrize("dtype", [F32, BF16, F16])
@pytest.mark.parametrize("c,size", [(16, (8, 12, 16)), (32, (4, 6, 10)), (48, (2, 2, 2)), (32, (16, 20, 12))])
def test_encoder_tail_with_upsampling_inside_vs_torch(dtype, c, size):
    """hdf_op_enc_tail_up: ds = relu(y * s + t) + Upsample(x2, trilinear)(relu(low * ls + lt)), MaxPool3d(2) of the stored
    ds (HDenseFormer.py:168-175,237-243): the level-0 encoder tail with at3 evaluated inside.  Edge voxels use torch's
    align_corners=False rule (sizes down to a single low-resolution voxel per axis); ties and the arg-max bytes as in
    test_encoder_tail_vs_torch."""
    n = 2
    lo = tuple(v // 2 for v in size)
    y, low = _mk((n, c) + size, 41), _mk((n, c) + lo, 42)
    y[:, :, :1] = -1.0
    scale, shift = _mk((n, c), 43) * 0.5 + 1.0, _mk((n, c), 44) * 0.3
    lscale, lshift = _mk((n, c), 45) * 0.5 + 1.0, _mk((n, c), 46) * 0.3
    scale[:, 0] = -scale[:, 0]
    bc = lambda v: v[:, :, None, None, None]
    up = F.interpolate(torch.relu(rnd(low, dtype) * bc(lscale) + bc(lshift)), scale_factor=2, mode="trilinear",
                       align_corners=False)
    ds_ref = rnd(torch.relu(rnd(y, dtype) * bc(scale) + bc(shift)) + up, dtype)
    y_cl, low_cl = to_cl(y, dtype), to_cl(low, dtype)
    ds = torch.empty_like(y_cl)
    po = torch.empty((n,) + lo + (c,), dtype=y_cl.dtype, device=DEV)
    idx = torch.empty(po.shape, dtype=torch.uint8, device=DEV)
    dev = [t.to(DEV).contiguous() for t in (scale, shift, lscale, lshift)]
    check(lib().hdf_op_enc_tail_up(dtype, ptr(y_cl), c, ptr(dev[0]), ptr(dev[1]), ptr(low_cl), c, ptr(dev[2]), ptr(dev[3]),
                                   ptr(ds), c, ptr(po), c, ptr(idx), n, c, *lo, st()), "enc_tail_up")
    torch.cuda.synchronize()
    got_ds, got_po = from_cl(ds), from_cl(po)
    assert rel_err(got_ds, ds_ref) < TOL[dtype]
    po2, idx2 = F.max_pool3d(got_ds, 2, return_indices=True)
    assert bool((got_po == po2).all())
    k = idx.permute(0, 4, 1, 2, 3).cpu().long()
    od, oh, ow = torch.meshgrid(*[torch.arange(v) for v in lo], indexing="ij")
    flat = ((2 * od + (k >> 2)) * size[1] + 2 * oh + ((k >> 1) & 1)) * size[2] + 2 * ow + (k & 1)
    assert bool((flat == idx2).all())


@pytest.mark.parametrize("dtype", [F32, BF16, F16])
@pytest.mark.parametrize("n,c,size", [(2, 32, (8, 12, 16)), (1, 64, (20, 16, 24)), (2, 16, (32, 32, 36))])
def test_instance_norm_relu_backward(dtype, n, c, size):
    """hdf_op_in_bwd (reduce + finalize + apply) vs autograd of relu(InstanceNorm3d(affine)(y)) (HDenseFormer.py:152-158)."""
    y = rnd(_mk((n, c) + size, 21), dtype).requires_grad_(True)
    gamma, beta = (_mk((c,), 22) * 0.3 + 1.0).requires_grad_(True), (_mk((c,), 23) * 0.2).requires_grad_(True)
    da = rnd(_mk((n, c) + size, 24), dtype)
    out = torch.relu(F.instance_norm(y, weight=gamma, bias=beta, eps=1e-5))
    out.backward(da)
    vox = size[0] * size[1] * size[2]
    yd = y.detach()
    mean = yd.mean((2, 3, 4))
    rstd = (yd.var((2, 3, 4), unbiased=False) + 1e-5).rsqrt()
    scale = (gamma.detach()[None] * rstd).contiguous()
    shift = (beta.detach()[None] - mean * scale).contiguous()
    dy = torch.empty((n,) + size + (c,), dtype=to_cl(yd, dtype).dtype, device=DEV)
    dg, db = torch.zeros(c, device=DEV), torch.zeros(c, device=DEV)
    ws = torch.empty(lib().hdf_op_in_bwd_workspace_floats(n, c, vox), device=DEV)
    dev = [t.to(DEV).contiguous() for t in (scale, shift, mean, rstd, gamma.detach())]
    da_cl, y_cl = to_cl(da, dtype), to_cl(yd, dtype)          # keep the device tensors alive across the call
    check(lib().hdf_op_in_bwd(dtype, ptr(da_cl), c, ptr(y_cl), c, ptr(dev[0]), ptr(dev[1]),
                              ptr(dev[2]), ptr(dev[3]), ptr(dev[4]), ptr(dy), c, ptr(dg), ptr(db), n, c, vox, ptr(ws),
                              st()), "in_bwd")
    torch.cuda.synchronize()
    tol = TOL[dtype] * (1.5 if dtype == BF16 else 20 if dtype == F32 else 3)   # fp32: cancellation in g - mean(g) - xhat*mean(g*xhat)
    assert rel_err(from_cl(dy), y.grad) < tol
    assert rel_err(dg.cpu(), gamma.grad) < 2e-3 and rel_err(db.cpu(), beta.grad) < 2e-3


@pytest.mark.parametrize("dtype", [BF16, F16])
@pytest.mark.parametrize("xf", [False, True])
@pytest.mark.parametrize("n,cin,cout,size", [(2, 32, 32, (16, 16, 24)),     # whole tiles, interior + border passes
                                            (1, 64, 32, (20, 18, 26)),     # ragged extents, two large-channel blocks
                                            (2, 32, 48, (12, 17, 9)),      # partial small-channel block (n_filters 48)
                                            (3, 48, 64, (8, 8, 16))])      # odd batch, two small-channel blocks
def test_instance_norm_backward_inside_the_weight_gradient(dtype, xf, n, cin, cout, size):
    """hdf_op_in_bwd_wgrad (conv_wgrad2_kernel<., ., true>: the second InstanceNorm-backward pass applied while the weight
    gradient stages d(activation), dy written as its by-product) against the unfused pair hdf_op_in_bwd + hdf_op_conv3d_wgrad
    on the same operands: dy and dw BIT FOR BIT; and dw against torch autograd of
    conv3d(relu(IN(.))) -> relu(IN(y)) loosely (the unfused ops carry the tight comparison).  Reference:
    models/HDenseFormer.py:148-159 backward."""
    vox = size[0] * size[1] * size[2]
    y = rnd(_mk((n, cout) + size, 51), dtype)
    da = rnd(_mk((n, cout) + size, 52), dtype)
    x = rnd(_mk((n, cin) + size, 53), dtype)
    gamma, beta = _mk((cout,), 54) * 0.3 + 1.0, _mk((cout,), 55) * 0.2
    mean = y.mean((2, 3, 4))
    rstd = (y.var((2, 3, 4), unbiased=False) + 1e-5).rsqrt()
    scale = (gamma[None] * rstd).contiguous()
    shift = (beta[None] - mean * scale).contiguous()
    xs, xh = (_mk((n, cin), 56) * 0.5 + 1.0, _mk((n, cin), 57) * 0.3) if xf else (None, None)
    dev = [t.to(DEV).contiguous() for t in (scale, shift, mean, rstd, gamma)]
    da_cl, y_cl, x_cl = to_cl(da, dtype), to_cl(y, dtype), to_cl(x, dtype)
    xs_d, xh_d = (xs.to(DEV), xh.to(DEV)) if xf else (None, None)
    ws = torch.empty(lib().hdf_op_in_bwd_workspace_floats(n, cout, vox), device=DEV)
    wsb = lib().hdf_op_wgrad_workspace_bytes(1, n, *size, cout, cin)
    wws = torch.empty(wsb, dtype=torch.uint8, device=DEV)

    def run(fused):
        dy = torch.full((n,) + size + (cout,), 7.0, dtype=y_cl.dtype, device=DEV)
        dg, db = torch.zeros(cout, device=DEV), torch.zeros(cout, device=DEV)
        dw = torch.zeros((cout, cin, 27), dtype=torch.float32, device=DEV)
        if fused:
            check(lib().hdf_op_in_bwd_wgrad(dtype, ptr(da_cl), cout, ptr(y_cl), cout, ptr(dev[0]), ptr(dev[1]), ptr(dev[2]),
                                            ptr(dev[3]), ptr(dev[4]), ptr(dy), cout, ptr(dg), ptr(db), ptr(x_cl), cin, cin,
                                            ptr(xs_d), ptr(xh_d), 1 if xf else 0, n, cout, *size, ptr(dw), ptr(ws),
                                            ptr(wws), wsb, st()), "in_bwd_wgrad")
        else:
            check(lib().hdf_op_in_bwd(dtype, ptr(da_cl), cout, ptr(y_cl), cout, ptr(dev[0]), ptr(dev[1]), ptr(dev[2]),
                                      ptr(dev[3]), ptr(dev[4]), ptr(dy), cout, ptr(dg), ptr(db), n, cout, vox, ptr(ws),
                                      st()), "in_bwd")
            check(lib().hdf_op_conv3d_wgrad(dtype, 1, ptr(dy), cout, cout, ptr(x_cl), cin, cin, n, *size, None, None, 0,
                                            ptr(xs_d), ptr(xh_d), 1 if xf else 0, ptr(dw), cout, cin, 0, ptr(wws), wsb,
                                            st()), "wgrad")
        torch.cuda.synchronize()
        return dy, dg, db, dw

    dy0, dg0, db0, dw0 = run(False)
    dy1, dg1, db1, dw1 = run(True)
    assert torch.equal(dy0.view(torch.int16), dy1.view(torch.int16))
    assert torch.equal(dw0, dw1)
    # (dgamma / dbeta are summed over the samples with float atomics: the same reduce + finalize launches in both runs)
    assert rel_err(dg1.cpu(), dg0.cpu()) < 1e-6 and rel_err(db1.cpu(), db0.cpu()) < 1e-6
    # and against autograd (storage-rounded operands, fp32 arithmetic)
    yt = y.clone().requires_grad_(True)
    act = torch.relu(F.instance_norm(yt, weight=gamma, bias=beta, eps=1e-5))
    act.backward(da)
    xin = torch.relu(x * xs[:, :, None, None, None] + xh[:, :, None, None, None]) if xf else x
    w = torch.zeros(cout, cin, 3, 3, 3, requires_grad=True)
    F.conv3d(rnd(xin, dtype), w, None, padding=1).backward(rnd(yt.grad, dtype))
    assert rel_err(dw1.cpu().view(cout, cin, 3, 3, 3), w.grad) < 3 * TOL[dtype]


@pytest.mark.parametrize("dtype,c", [(F32, 16), (BF16, 32), (F16, 32)])
@pytest.mark.parametrize("n,size", [(2, (32, 32, 32)), (1, (33, 38, 41))])
def test_upsample_bwd_large_pitched_vs_autograd(dtype, c, n, size):
    """larger, odd extents; the gradient is read from a channel slice of a wider buffer (as the UpConv chain reads it
    from the concat gradient)."""
    xa = _mk((n, c) + size, 31).requires_grad_(True)
    ref = F.interpolate(xa, scale_factor=2, mode="trilinear", align_corners=False)
    g = _mk(tuple(ref.shape), 32)
    ref.backward(rnd(g, dtype))
    gcl = to_cl(g, dtype)
    wide = torch.zeros(gcl.shape[:-1] + (2 * c,), dtype=gcl.dtype, device=DEV)
    wide[..., c:] = gcl
    view = wide.view(-1)[c:]
    dlo = torch.empty((n,) + size + (c,), dtype=gcl.dtype, device=DEV)
    check(lib().hdf_op_upsample_bwd(dtype, ptr(view), 2 * c, ptr(dlo), c, n, c, *size, st()), "upb")
    torch.cuda.synchronize()
    assert rel_err(from_cl(dlo), xa.grad) < TOL[dtype]


@pytest.mark.parametrize("dtype", [BF16])
def test_transposed_conv_kernels_at_benchmark_size_are_reproducible(dtype):
    """The three LDS-DMA kernels of the top-level ConvTranspose3d at the benchmarked extent (64 <-> 32 channels, 64^3 ->
    128^3, batch 2: 32 tiles per workgroup, every pipeline stage in steady state): two launches on the same inputs are
    bitwise equal (a missed wait on a DMA or a buffer reused too early shows up here), and a corner of the forward and
    of the data gradient equals torch's result on the sub-volume it depends on."""
    n, cl, ch, s = 2, 64, 32, 64
    g = torch.Generator().manual_seed(77)
    lo = (torch.randn(n, s, s, s, cl, generator=g) * 0.5).to(TDT[dtype]).to(DEV)
    hi = (torch.randn(n, 2 * s, 2 * s, 2 * s, ch, generator=g) * 0.5).to(TDT[dtype]).to(DEV)
    sc = (torch.rand(n, cl, generator=g) + 0.5).to(DEV)
    sh = (torch.randn(n, cl, generator=g) * 0.1).to(DEV)
    w_t = torch.randn(cl, ch, 3, 3, 3, generator=g) * (cl * 27 / 8) ** -0.5
    # forward
    wp = pack_w(w_t, dtype, ch, cl, rup(ch, 32), cl, 27, ch * 27, 0)
    outs = []
    for _ in range(2):
        out = torch.zeros(n, 2 * s, 2 * s, 2 * s, ch, dtype=lo.dtype, device=DEV)
        check(lib().hdf_op_conv3d(dtype, 2, ptr(lo), cl, cl, n, s, s, s, ptr(wp), None, ptr(sc), ptr(sh), 1, ptr(out), ch, ch,
                                  None, 0, st()), "convt")
        outs.append(out)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1])
    xin = torch.relu(lo[:, :12, :12, :12].float() * sc[:, None, None, None, :] + sh[:, None, None, None, :]).to(lo.dtype).float()
    ref = F.conv_transpose3d(xin.permute(0, 4, 1, 2, 3), rnd(w_t, dtype).to(DEV), None, stride=2, padding=1, output_padding=1)
    got = outs[0][:, :20, :20, :20].float().permute(0, 4, 1, 2, 3)      # outputs that depend on the 12^3 corner only
    assert rel_err(got.cpu(), ref[:, :, :20, :20, :20].cpu()) < TOL[dtype]
    # data gradient (stride-2 gather conv, 32 -> 64)
    w_g = torch.randn(cl, ch, 3, 3, 3, generator=g) * (ch * 27) ** -0.5
    wpg = pack_w(w_g, dtype, cl, ch, rup(cl, 32), ch, ch * 27, 27, 0)
    gs = []
    for _ in range(2):
        o = torch.zeros(n, s, s, s, cl, dtype=lo.dtype, device=DEV)
        check(lib().hdf_op_conv3d(dtype, 1, ptr(hi), ch, ch, n, 2 * s, 2 * s, 2 * s, ptr(wpg), None, None, None, 0, ptr(o), cl, cl,
                                  None, 0, st()), "gather")
        gs.append(o)
    torch.cuda.synchronize()
    assert torch.equal(gs[0], gs[1])
    refg = F.conv3d(hi[:, :25, :25, :25].float().permute(0, 4, 1, 2, 3), rnd(w_g, dtype).to(DEV), None, stride=2, padding=1)
    assert rel_err(gs[0][:, :12, :12, :12].float().permute(0, 4, 1, 2, 3).cpu(), refg[:, :, :12, :12, :12].cpu()) < TOL[dtype]
    # weight gradient
    wsb = lib().hdf_op_wgrad_workspace_bytes(2, n, s, s, s, cl, ch)
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    dws = []
    for _ in range(2):
        dw = torch.zeros(cl, ch, 27, device=DEV)
        check(lib().hdf_op_conv3d_wgrad(dtype, 2, ptr(lo), cl, cl, ptr(hi), ch, ch, n, s, s, s, ptr(sc), ptr(sh), 1, None, None, 0,
                                        ptr(dw), cl, ch, 0, ptr(ws), wsb, st()), "wgrad2")
        dws.append(dw)
    torch.cuda.synchronize()
    assert torch.equal(dws[0], dws[1])
    assert torch.isfinite(dws[0]).all() and float(dws[0].abs().max()) > 0
