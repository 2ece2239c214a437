"""Where does the persistent backward kernel differ from the launch chain?  Tapes per layer and segment, last layer first.
usage: python tools/chain_diff_bwd.py <case index of tests/test_gpu_chain.py::CASES>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "h-denseformer_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import test_gpu_chain as tc  # noqa: E402

case = tc.CASES[int(sys.argv[1]) if len(sys.argv) > 1 else 0]
ref = tc._backward(case, chain=False)
got = tc._backward(case, chain=True)
cin, ncls, nf, image, depth, batch, dtype = case
N = (image[0] // 16) ** 3
rows = cin * batch * N
nl = (depth // 4) * 4
DM = 4 * nf
DMF = DM + 128
nseq = cin * batch
print("sync fwd", got["sync"][::32][:nseq + 1].tolist(), "bwd", got["sync"][(1 << 17)::32][:nseq + 1].tolist())
segs = [("DQ", 0, 96), ("T", 96, 32), ("DH0", 128, 32), ("P1.dg", 160, 32), ("P1.f", 192, 64), ("P1.dz", 256, 64),
        ("P1.u", 320, 32), ("P0.dg", 352, 32), ("P0.f", 384, 64), ("P0.dz", 448, 64), ("P0.u", 512, 32), ("DGO", 544, 32)]


def cmp(x, y, name):
    nd = x.view(torch.int32) != y.view(torch.int32)
    if nd.any():
        r = nd.any(dim=1).nonzero().flatten()
        return "%s: %d words rows %s max|d| %.2e" % (name, int(nd.sum()), r[:5].tolist(), float((x - y).abs().max()))
    return None


for L in range(nl - 1, -1, -1):
    if L % 4 == 3:
        b = L // 4
        a = got["otape"][b * rows * DMF:(b + 1) * rows * DMF]
        c = ref["otape"][b * rows * DMF:(b + 1) * rows * DMF]
        out = []
        for name, o, w in [("do", 0, DM), ("f", DM, 64), ("dz", DM + 64, 64)]:
            r = cmp(a[rows * o: rows * (o + w)].view(rows, w), c[rows * o: rows * (o + w)].view(rows, w), name)
            if r:
                out.append(r)
        print("block", b, "out tape:", "; ".join(out) if out else "identical")
    a = got["tape"][L * rows * 576:(L + 1) * rows * 576]
    c = ref["tape"][L * rows * 576:(L + 1) * rows * 576]
    out = []
    for name, o, w in segs:
        r = cmp(a[rows * o: rows * (o + w)].view(rows, w), c[rows * o: rows * (o + w)].view(rows, w), name)
        if r:
            out.append(r)
    print("layer", L, "; ".join(out) if out else "identical")
r = cmp(got["dF"], ref["dF"], "dF[:, :DM]")
print(r or "dF identical")
