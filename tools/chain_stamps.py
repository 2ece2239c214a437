"""Phase timing inside the persistent transformer kernels (a -DCHAIN_DBG_STAMPS build: python h-denseformer_amd/build.py
--name libhdf_hip_stamps -DCHAIN_DBG_STAMPS; run with HDF_LIB_PATH=.../libhdf_hip_stamps.so).  Lane 0 of every workgroup
stamps the 100 MHz real-time counter at the phase boundaries of every layer; this prints the median / max over workgroups
of each phase, averaged over the layers, in microseconds."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "h-denseformer_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import test_gpu_chain as tc  # noqa: E402

case = tc.CASES[int(sys.argv[1]) if len(sys.argv) > 1 else 3]
which = sys.argv[2] if len(sys.argv) > 2 else "fwd"
cin, ncls, nf, image, depth, batch, dtype = case
N = (image[0] // 16) ** 3
nwg = cin * batch * ((N + 15) // 16)
nl = (depth // 4) * 4
if which == "bwd":
    for rep in range(2):
        got = tc._backward(case, chain=True)
    sync = got["sync"]
    st = sync[(1 << 19):(1 << 19) + nwg * 32 * 16 * 2].view(torch.int64).view(nwg, 32, 16).cpu().double() / 100.0
    # backward stamps are indexed by L + 1 (L = nl - 1 .. -1): row r = token phase that ends with the attention of layer r - 1
    tot = st[:, 0, 2] - st[:, nl, 0]
    print("kernel span per workgroup: median %.1f us, max %.1f us" % (tot.median(), tot.max()))
    PHB = [("requests issued -> barrier", 0, 1), ("PREB", 1, 2), ("OUTB (block boundaries only)", 2, 3), ("POSTB", 3, 4),
           ("  POSTB pass 1: LN keep + mask", 3, 9), ("  pass 1: two GEMMs + GELU epilogue", 9, 10), ("  pass 1: du GEMM (+ side outputs)", 10, 11),
           ("  pass 1: reduce", 11, 12), ("  pass 1: LN backward", 12, 13), ("  pass 0 (whole)", 13, 14),
           ("  to_out: dg, wgrad operands, dO GEMM", 14, 15), ("  to_out: reduce, delta", 15, 4),
           ("publish", 4, 5), ("poll", 5, 6), ("attention dQ half", 6, 7), ("attention dK/dV half", 7, 8)]
    R = torch.arange(2, nl - 1)
    inner = R[((R - 1) % 4) != 3]
    for name, a, b in PHB:
        Rs = R[((R - 1) % 4) == 3] if "OUTB" in name else inner
        dd = st[:, Rs, b] - st[:, Rs, a]
        print("%-42s median %6.2f  mean %6.2f  max-over-wg (mean over layers) %6.2f" % (name, dd.median(), dd.mean(), dd.max(dim=0).values.mean()))
    nxt = st[:, R - 1, 0] - st[:, R, 8]
    print("%-42s median %6.2f" % ("attention end -> next phase's requests", nxt.median()))
    per = st[:, R - 1, 0] - st[:, R, 0]
    print("per layer: median %.2f us, mean %.2f us" % (per.median(), per.mean()))
    sys.exit(0)
for rep in range(3):
    got, keep = tc._forward(case, chain=True)
sync = got["sync"]
st = sync[(1 << 18):(1 << 18) + nwg * 32 * 16 * 2].view(torch.int64).view(nwg, 32, 16).cpu().double() / 100.0   # us
tot = (st[:, nl, 1] - st[:, 0, 0])
print("kernel span per workgroup: median %.1f us, max %.1f us; launch skew of stamp 0: %.1f us" %
      (tot.median(), tot.max(), st[:, 0, 0].max() - st[:, 0, 0].min()))
# (name, from stamp, to stamp, to-stamp is of the NEXT layer)
PH = [("barrier at the loop top", 8, 0, 0), ("POST to_out", 0, 9, 0), ("POST ff pass 0", 9, 10, 0), ("POST ff pass 1", 10, 13, 0),
      ("OUT (block boundaries only)", 13, 1, 0), ("PRE Linear0 + reduce", 1, 11, 0), ("PRE LN1", 11, 12, 0),
      ("PRE to_qkv -> LDS", 12, 2, 0), ("publish (stores, drain, barrier, add)", 2, 3, 0),
      ("request POST weights + poll", 3, 4, 0), ("first chunk staged", 4, 5, 0), ("attention chunks", 5, 6, 0),
      ("merge + ob / lse stores", 6, 7, 0), ("request PRE weights (next layer)", 7, 8, 1)]
L = torch.arange(1, nl - 1)
inner = L[(L % 4) != 0]
for name, a, b, nxt in PH:
    Ls = inner if "OUT" not in name and a != 13 else L[(L % 4) == 0]
    d = st[:, Ls + nxt, b] - st[:, Ls, a]
    print("%-42s median %6.2f  mean %6.2f  max-over-wg (mean over layers) %6.2f" %
          (name, d.median(), d.mean(), d.max(dim=0).values.mean()))
per_layer = (st[:, L + 1, 0] - st[:, L, 0])
print("per layer: median %.2f us, mean %.2f us" % (per_layer.median(), per_layer.mean()))
