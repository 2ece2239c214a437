"""SURVEY 5.2 / VERDICT r03 #7a: the library writes only where it was told to.  Forward + backward through the C ABI with
the workspace, the four outputs and the flat gradient buffer each embedded in a larger allocation whose guard bands
(1 MiB on either side, patterned) must come back untouched -- at the benchmark geometry (BASELINE configs[1]) and at the
geometries of configs[3] (2 x 144^3, 3 classes) and configs[4] (n_filters 48, 4 x 160^3), bf16 storage, train mode.
The reference has nothing of the kind (test.py:1-40 is a shape check); the kernels here carve ~4 GB arenas by hand."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "h-denseformer_amd"), os.path.dirname(os.path.abspath(__file__))):
    sys.path.insert(0, p)
from hdf_rt._lib import BF16, check, lib, ptr  # noqa: E402
from hdf_rt.runtime import Plan  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GUARD = 1 << 20          # bytes on either side


class Guarded:
    """`nbytes` of payload between two guard bands filled with a pattern the kernels would have to reproduce by chance."""

    def __init__(self, nbytes, tag):
        self.n = (nbytes + 255) // 256 * 256
        self.buf = torch.empty(self.n + 2 * GUARD, dtype=torch.uint8, device=DEV)
        self.pattern = (torch.arange(GUARD, device=DEV, dtype=torch.int64) * 131 + tag).to(torch.uint8)
        self.buf[:GUARD] = self.pattern
        self.buf[GUARD + self.n:] = self.pattern
        self.payload = self.buf[GUARD:GUARD + self.n]
        self.payload.fill_(0x5A)

    def intact(self):
        lo = bool((self.buf[:GUARD] == self.pattern).all())
        hi = bool((self.buf[GUARD + self.n:] == self.pattern).all())
        return lo, hi


@pytest.mark.parametrize("cfg,batch", [((4, 4, 32, (128, 128, 128), 24), 2),      # BASELINE configs[1]: what BENCH times
                                       ((2, 3, 32, (144, 144, 144), 24), 1),      # configs[3] geometry (odd token grid 9^3)
                                       ((4, 4, 48, (160, 160, 160), 24), 1)])     # configs[4] geometry (96-byte rows)
def test_forward_backward_stay_inside_their_buffers(cfg, batch):
    in_ch, n_cls, nf, size, depth = cfg
    plan = Plan(in_ch, n_cls, nf, size, depth, BF16)
    g = torch.Generator().manual_seed(11)
    flat = (torch.randn(plan.param_floats, generator=g) * 0.02).to(DEV)
    # InstanceNorm / LayerNorm gains near one so that activations stay finite through 20+ layers
    for name, off, numel, _shape in plan.table:
        if name.endswith("norm.weight"):
            flat[off:off + numel] = 1.0
    x = torch.rand((batch, in_ch) + tuple(size), generator=g).to(DEV)
    ws = Guarded(plan.workspace_bytes(batch), 1)
    outs = [Guarded(batch * n_cls * (size[0] >> i) * (size[1] >> i) * (size[2] >> i) * 2, 2 + i) for i in range(4)]
    grads = Guarded(plan.param_floats * 4, 7)
    st = torch.cuda.current_stream().cuda_stream
    check(lib().hdf_forward(plan.h, ptr(x), ptr(flat), ptr(ws.payload), ws.n, ptr(outs[0].payload), ptr(outs[1].payload),
                            ptr(outs[2].payload), ptr(outs[3].payload), batch, 1, 12345, st), "hdf_forward")
    torch.cuda.synchronize()
    for name, gb in [("workspace", ws)] + [(f"out{i}", o) for i, o in enumerate(outs)]:
        assert gb.intact() == (True, True), f"forward wrote outside {name}"
    # finite logits (the payload is a view: whole tensors, storage dtype)
    for i, o in enumerate(outs):
        n = batch * n_cls * (size[0] >> i) * (size[1] >> i) * (size[2] >> i)
        assert bool(torch.isfinite(o.payload[:2 * n].view(torch.bfloat16).float()).all()), f"out{i} not finite"
    douts = [(torch.randn(batch * n_cls * (size[0] >> i) * (size[1] >> i) * (size[2] >> i), generator=g) * 1e-3)
             .to(torch.bfloat16).to(DEV) for i in range(4)]
    check(lib().hdf_backward(plan.h, ptr(x), ptr(flat), ptr(ws.payload), ws.n, ptr(douts[0]), ptr(douts[1]), ptr(douts[2]),
                             ptr(douts[3]), ptr(grads.payload), batch, st), "hdf_backward")
    torch.cuda.synchronize()
    for name, gb in [("workspace", ws), ("gradient buffer", grads)] + [(f"out{i}", o) for i, o in enumerate(outs)]:
        assert gb.intact() == (True, True), f"backward wrote outside {name}"
    gflat = grads.payload[:plan.param_floats * 4].view(torch.float32)
    assert bool(torch.isfinite(gflat).all())
    assert float(gflat.abs().max()) > 0.0
