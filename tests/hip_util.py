"""Helpers for the GPU parity tests: call the C ABI (through hdf_rt._lib) on torch tensors."""
import torch

from hdf_rt._lib import BF16, F16, F32, check, lib, ptr

TDT = {F32: torch.float32, BF16: torch.bfloat16, F16: torch.float16}
DEV = "cuda:0"


def st():
    return torch.cuda.current_stream().cuda_stream


def to_cl(x, dtype, cp=None):
    """[N,C,D,H,W] fp32 (cpu or gpu) -> channels-last storage tensor [N,D,H,W,CP] on the GPU."""
    n, c = x.shape[:2]
    cp = cp or c
    out = torch.zeros((n,) + tuple(x.shape[2:]) + (cp,), dtype=TDT[dtype], device=DEV)
    out[..., :c] = x.to(DEV).permute(0, 2, 3, 4, 1).to(TDT[dtype])
    return out.contiguous()


def from_cl(t):
    """channels-last [N,D,H,W,C] storage -> [N,C,D,H,W] fp32 on the CPU"""
    return t.float().permute(0, 4, 1, 2, 3).contiguous().cpu()


def rnd(x, dtype):
    """round a float tensor through the storage dtype"""
    return x.to(TDT[dtype]).float()


def pack_w(w, dtype, O, I, OP, IP, so, si, flip):
    dst = torch.empty(27 * OP * IP, dtype=TDT[dtype], device=DEV)
    wg = w.contiguous().to(DEV)
    check(lib().hdf_op_pack_weights(dtype, ptr(wg), ptr(dst), O, I, OP, IP, so, si, flip, st()), "pack")
    return dst


def rup(a, b):
    return (a + b - 1) // b * b


def conv3d(dtype, mode, x_cl, cin, w_packed, cout, bias=None, scale=None, shift=None, relu=0, stats=False,
           out=None, out_pitch=None, accumulate=0):
    n, d, h, w = x_cl.shape[:4]
    pitch = x_cl.shape[4]
    if mode == 0:
        od, oh, ow = d, h, w
    elif mode == 1:
        od, oh, ow = d // 2, h // 2, w // 2
    else:
        od, oh, ow = 2 * d, 2 * h, 2 * w
    if d == 1:
        od = 1          # depth 1 selects the 2-D operator: the depth axis is never strided
    if out is None:
        out = torch.zeros((n, od, oh, ow, cout), dtype=x_cl.dtype, device=DEV)
        out_pitch = cout
    part = None
    if stats:
        tiles = lib().hdf_op_conv3d_stat_tiles(dtype, cin, od, oh, ow)
        part = torch.zeros((n * tiles, rup(cout, 32), 2), dtype=torch.float32, device=DEV)
    check(lib().hdf_op_conv3d(dtype, mode, ptr(x_cl), pitch, cin, n, d, h, w, ptr(w_packed), ptr(bias), ptr(scale),
                              ptr(shift), relu, ptr(out), out_pitch, cout, ptr(part), accumulate, st()), "conv3d")
    return out, part


def rel_err(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def rel_l2(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))
