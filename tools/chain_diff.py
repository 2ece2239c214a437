"""Where do the persistent transformer kernels differ from the launch chain?  Per layer and saved segment: differing words.
usage: python tools/chain_diff.py <case index of tests/test_gpu_chain.py::CASES>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "h-denseformer_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import test_gpu_chain as tc  # noqa: E402

case = tc.CASES[int(sys.argv[1]) if len(sys.argv) > 1 else 1]
ref, _ = tc._forward(case, chain=False)
got, _ = tc._forward(case, chain=True)
cin, ncls, nf, image, depth, batch, dtype = case
N = (image[0] // 16) ** 3
rows = cin * batch * N
nl = (depth // 4) * 4
segs = [("h0", 0, 32), ("qkv", 32, 96), ("ob", 128, 32), ("lse", 160, 8), ("h1", 168, 32), ("h2", 200, 32)]
print("sync", got["sync"][:: 32][: cin * batch + 1].tolist())
for L in range(nl):
    a = got["save"][L * rows * 232:(L + 1) * rows * 232]
    b = ref["save"][L * rows * 232:(L + 1) * rows * 232]
    out = []
    for name, o, w in segs:
        x = a[rows * o: rows * (o + w)].view(rows, w)
        y = b[rows * o: rows * (o + w)].view(rows, w)
        nd = (x.view(torch.int32) != y.view(torch.int32))
        if nd.any():
            r = nd.any(dim=1).nonzero().flatten()
            out.append("%s: %d words, rows %s.., max|d| %.2e" % (name, int(nd.sum()), r[:6].tolist(), float((x - y).abs().max())))
    print("layer", L, "; ".join(out) if out else "identical")
DMF = 4 * nf + 128
F = got["F"].view(-1, rows, DMF)
G = ref["F"].view(-1, rows, DMF)
for b in range(F.shape[0]):
    nd = F[b].view(torch.int32) != G[b].view(torch.int32)
    print("F block", b, int(nd.sum()), "cols", nd.any(dim=0).nonzero().flatten()[:8].tolist(), "rows", nd.any(dim=1).nonzero().flatten()[:8].tolist())
