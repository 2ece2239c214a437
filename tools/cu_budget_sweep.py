"""Persistent conv / wgrad kernels at different workgroup counts (hdf_set_cu_budget) on the full chip: under the
package power cap fewer active CUs can clock higher.  Interleaved rounds in one process (same device, same data)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "h-denseformer_amd")):
    sys.path.insert(0, p)
import torch

from hdf_rt._lib import BF16, check, lib, ptr

dev = "cuda:0"
st = torch.cuda.current_stream().cuda_stream
budgets = [int(b) for b in os.environ.get("BUDGETS", "160,192,208,224,240,256").split(",")]
shapes = [("conv", 64, 32, 128, 2), ("conv", 32, 32, 128, 2), ("conv", 32, 64, 128, 2), ("conv", 64, 64, 64, 2),
          ("wgrad", 64, 32, 128, 2), ("wgrad", 32, 32, 128, 2), ("wgrad", 64, 64, 64, 2)]
rounds, reps = 3, 40


def make(op, cin, cout, s, n):
    x = torch.randn(n, s, s, s, cin, device=dev).to(torch.bfloat16)
    if op == "conv":
        coutp = (cout + 31) // 32 * 32
        w = (torch.randn(27 * coutp * cin, device=dev) * 0.02).to(torch.bfloat16)
        out = torch.empty(n, s, s, s, cout, device=dev, dtype=torch.bfloat16)
        tiles = lib().hdf_op_conv3d_stat_tiles(BF16, cin, s, s, s)
        part = torch.empty(n * tiles * coutp * 2, device=dev)

        def launch():
            check(lib().hdf_op_conv3d(BF16, 0, ptr(x), cin, cin, n, s, s, s, ptr(w), None, None, None, 0, ptr(out), cout,
                                      cout, ptr(part), 0, st), "conv")
        return launch, (x, w, out, part)
    dy = torch.randn(n, s, s, s, cout, device=dev).to(torch.bfloat16)
    wsb = lib().hdf_op_wgrad_workspace_bytes(1, n, s, s, s, cout, cin)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    dw = torch.zeros(cout, cin, 27, device=dev)

    def launch():
        check(lib().hdf_op_conv3d_wgrad(BF16, 1, ptr(dy), cout, cout, ptr(x), cin, cin, n, s, s, s, None, None, 0, None,
                                        None, 0, ptr(dw), cout, cin, 0, ptr(ws), wsb, st), "wgrad")
    return launch, (x, dy, ws, dw)


e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for shp in shapes:
    launch, keep = make(*shp)
    res = {b: [] for b in budgets}
    for r in range(rounds):
        for b in budgets:
            check(lib().hdf_set_cu_budget(b), "budget")
            for _ in range(8):
                launch()
            e0.record()
            for _ in range(reps):
                launch()
            e1.record()
            torch.cuda.synchronize()
            res[b].append(e0.elapsed_time(e1) / reps * 1e3)
    check(lib().hdf_set_cu_budget(256), "budget")
    print(shp, {b: "%.1f" % min(v) for b, v in res.items()}, flush=True)
    del keep
