"""Micro-driver for profiling one conv / wgrad shape through the C ABI (used with rocprofv3 --pmc)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "h-denseformer_amd")):
    sys.path.insert(0, p)
import torch

from hdf_rt._lib import BF16, F32, check, lib, ptr

ap = argparse.ArgumentParser()
ap.add_argument("--op", default="conv")
ap.add_argument("--cin", type=int, default=64)
ap.add_argument("--cout", type=int, default=32)
ap.add_argument("--size", type=int, default=128)
ap.add_argument("--n", type=int, default=2)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--xf", type=int, default=0)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--wr", type=int, default=0, help="1: force conv_wr_kernel (hdf_op_conv3d_wr)")
ap.add_argument("--bs", type=int, default=0, help="1: the data-gradient form with the statistics epilogue (hdf_op_conv3d_bwd_stats)")
ap.add_argument("--cold", type=int, default=0, help="1: evict L2 / memory-side cache before every timed launch")
a = ap.parse_args()
dt = BF16 if a.dtype == "bf16" else F32
tdt = torch.bfloat16 if dt == BF16 else torch.float32
dev = "cuda:0"
n, cin, cout, s = a.n, a.cin, a.cout, a.size
st = torch.cuda.current_stream().cuda_stream
x = torch.randn(n, s, s, s, cin, device=dev).to(tdt)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
if a.op == "conv":
    coutp = (cout + 31) // 32 * 32
    w = (torch.randn(27 * coutp * cin, device=dev) * 0.02).to(tdt)
    out = torch.empty(n, s, s, s, cout, device=dev, dtype=tdt)
    tiles = lib().hdf_op_conv3d_stat_tiles(dt, cin, s, s, s)
    part = torch.empty(n * tiles * coutp * 2, device=dev)
    sc = torch.rand(n, cin, device=dev) + 0.5 if a.xf else None
    sh = torch.randn(n, cin, device=dev) * 0.1 if a.xf else None

    if a.bs:
        ybs = torch.randn(n, s, s, s, cout, device=dev).to(tdt)
        bsv = [torch.rand(n * cout, device=dev) + 0.5, torch.randn(n * cout, device=dev) * 0.1,
               torch.randn(n * cout, device=dev) * 0.1, torch.rand(n * cout, device=dev) + 0.5]
        part = torch.empty(max(n * tiles, n * 512) * coutp * 2, device=dev)

    def launch():
        if a.bs:
            check(lib().hdf_op_conv3d_bwd_stats(dt, ptr(x), cin, cin, n, s, s, s, ptr(w), ptr(out), cout, cout, ptr(ybs), cout,
                                                ptr(bsv[0]), ptr(bsv[1]), ptr(bsv[2]), ptr(bsv[3]), ptr(part), st), "conv_bs")
            return
        if a.wr:
            check(lib().hdf_op_conv3d_wr(dt, ptr(x), cin, cin, n, s, s, s, ptr(w), None, ptr(sc), ptr(sh), 1, ptr(out),
                                         cout, cout, ptr(part), 0, st), "conv_wr")
            return
        check(lib().hdf_op_conv3d(dt, 0, ptr(x), cin, cin, n, s, s, s, ptr(w), None, ptr(sc), ptr(sh), 1, ptr(out), cout,
                                  cout, ptr(part), 0, st), "conv")
    flops = 2.0 * 27 * cin * cout * s ** 3 * n
else:
    dy = torch.randn(n, s, s, s, cout, device=dev).to(tdt)
    wsb = lib().hdf_op_wgrad_workspace_bytes(1, n, s, s, s, cout, cin)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    dw = torch.zeros(cout, cin, 27, device=dev)

    def launch():
        check(lib().hdf_op_conv3d_wgrad(dt, 1, ptr(dy), cout, cout, ptr(x), cin, cin, n, s, s, s, None, None, 0, None,
                                        None, 0, ptr(dw), cout, cin, 0, ptr(ws), wsb, st), "wgrad")
    flops = 2.0 * 27 * cin * cout * s ** 3 * n
for _ in range(1 if a.cold else 15):      # clocks ramp over the first launches: time the settled state
    launch()
torch.cuda.synchronize()
if a.cold:
    junk = torch.empty(1 << 30, dtype=torch.uint8, device=dev)   # 1 GiB > L2 + 256 MB memory-side cache
    ms = 0.0
    for _ in range(a.reps):
        junk.add_(1)
        e0.record()
        launch()
        e1.record()
        torch.cuda.synchronize()
        ms += e0.elapsed_time(e1) / a.reps
else:
    e0.record()
    for _ in range(a.reps):
        launch()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.reps
if os.environ.get("WR_STAMPS") and a.op == "conv":
    # conv_wr.hip -DWR_DBG_STAMPS: [workgroup][wave 0/1][16]: cycles per section, total cycles, real time in 10 ns
    t = part[:256 * 2 * 16].view(256, 2, 16).double().cpu()
    names = ["mfma-phase", "barrier1", "xch-write", "barrier2", "xch-read", "barrier3", "epilogue", "step/top"]
    for w in range(2):
        m = t[:, w, :8].mean(0).tolist()
        tot = t[:, w, 8].mean().item()
        print(f"  wave {w}: total {int(tot)} cycles, clock {tot / (t[:, w, 9].mean().item() * 10):.2f} GHz; " +
              ", ".join(f"{n} {v / tot * 100:.1f}%" for n, v in zip(names, m)))
if os.environ.get("WS_STAMPS") and a.op == "conv":
    t = part[:256 * 8].view(256, 8).double().cpu()
    names = ["commit", "barrier1", "prefetch-issue", "mfma-loop", "barriers2+3", "epi:stage+stats", "epi:barrier+stores", "loop-top"]
    if not os.environ.get("HDF_WS_OLD"):
        names = ["xf-read", "mfma-phase", "barrier", "pass/tile-top", "epilogue", "-", "-", "-"]
    tot = t[:, :5].sum(1).mean().item()
    print("  per-WG cycles (mean over WGs):", {n: int(v) for n, v in zip(names[:5], t[:, :5].mean(0).tolist())}, "total", int(tot))
    if not os.environ.get("HDF_WS_OLD"):
        raw = part[:256 * 8].view(256, 8).view(torch.int32).cpu().long() & 0xFFFFFFFF
        t0, t1 = raw[:, 5], raw[:, 6]
        dur = ((t1 - t0) & 0xFFFFFFFF).double() / 100.0          # us (100 MHz counter)
        start = ((t0 - t0.min()) & 0xFFFFFFFF).double() / 100.0
        end = start + dur
        cyc = t[:, 7]
        q = lambda v: [round(float(x), 1) for x in torch.quantile(v, torch.tensor([0.0, 0.1, 0.5, 0.9, 1.0], dtype=torch.float64))]
        ent = part[2048:2048 + 256].view(torch.int32).cpu().long() & 0xFFFFFFFF
        setup = ((t0 - ent) & 0xFFFFFFFF).double() / 100.0
        print("  per-WG us from kernel entry to the first stamp (weight staging + per-lane setup):", q(setup),
              "; entry spread", q(((ent - ent.min()) & 0xFFFFFFFF).double() / 100.0),
              "; entry of the first WG to the end of the last: %.1f" % float((((t1 - ent.min()) & 0xFFFFFFFF).double() / 100.0).max()))
        print("  per-WG us: start", q(start), "duration", q(dur), "end", q(end))
        print("  in-kernel clock GHz (cycles / real time):", q(cyc / dur / 1e3))
        for x in range(8):
            sel = torch.arange(256) % 8 == x
            print(f"   blockIdx%8={x}: duration median {float(dur[sel].median()):.1f} max {float(dur[sel].max()):.1f} end max {float(end[sel].max()):.1f}")
if os.environ.get("WS_STAMPS") and a.op == "wgrad":
    per = 27 * ((cout + 31) // 32 * 32) * ((cin + 31) // 32 * 32)
    t = ws.view(torch.float32)[: 256 * per].view(256, per)[:, :8].double().cpu()
    names = ["top", "kloop-fast", "barrier", "-", "-", "kloop-slow", "-", "-"]
    print("  per-WG cycles (mean over groups):", {n: int(v) for n, v in zip(names, t.mean(0).tolist())}, "total", int(t.sum(1).mean().item()))
print(f"{a.op} {cin}->{cout} @{s}^3 n={n} {a.dtype} xf={a.xf} bs={a.bs}: {ms*1e3:.1f} us  {flops/ms/1e9:.1f} TFLOP/s")
