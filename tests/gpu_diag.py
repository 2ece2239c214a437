"""Ad-hoc GPU diagnostics (not a pytest file): gradient localisation against the oracle."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "h-denseformer_amd")):
    sys.path.insert(0, p)
import torch

from oracle import detgen
from oracle import hdf_oracle as orc


def rl2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def main():
    from loss.combine_loss import CEPlusDice, DeepSuperloss
    from models.HDenseFormer import HDenseFormer
    cfg = (2, 3, 16, (32, 32, 32), 8)
    sd = orc.det_model(*cfg)
    net = HDenseFormer(cfg[0], cfg[1], cfg[2], image_size=cfg[3], transformer_depth=cfg[4])
    net.load_state_dict(sd)
    net = net.to("cuda:0").eval()
    x = torch.from_numpy(detgen.det_input(2, cfg[0], cfg[3], tag="g1_tiny_eval"))
    onehot = torch.from_numpy(detgen.one_hot(detgen.det_labels(2, cfg[1], cfg[3], tag="g1_tiny_eval"), cfg[1]))
    crit = DeepSuperloss(criterion=CEPlusDice(weight=None, ignore_index=0))
    outs = net(x.cuda())
    for o in outs:
        o.retain_grad()
    loss = crit(outs, onehot.cuda())
    loss.backward()
    torch.cuda.synchronize()
    rt = net._last_rt

    ref = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    routs, inter = orc.forward(x, ref, None, want_intermediates=True)
    for v in inter.values():
        v.retain_grad()
    for o in routs:
        o.retain_grad()
    rloss = orc.deep_super_loss(routs, onehot)
    rloss.backward()
    print("loss", loss.item(), rloss.item())
    for i in range(4):
        print(f"dlogits{i} rel-l2 {rl2(outs[i].grad, routs[i].grad):.3e}")
    def read(name):
        # the gradient of cat_k is one buffer "g.cat{k}" or, when the halves are whole 32-channel blocks, the two
        # dense buffers "g.up{k}" (upconv half) and "g.ds{k-1}" (skip half)
        if name.startswith("g.cat"):
            k = int(name[-1])
            try:
                return rt.read_buffer(name)
            except Exception:
                return torch.cat([rt.read_buffer(f"g.up{k}"), rt.read_buffer(f"g.ds{k - 1}")], dim=1)
        return rt.read_buffer(name)

    pairs = [("g.cat1", "cat1"), ("g.cat2", "cat2"), ("g.cat3", "cat3"), ("g.attnout", "attnout"), ("g.attnall", "attnall")]
    for mine, theirs in pairs:
        a, b = read(mine), inter[theirs].grad
        print(f"{mine:10s} rel-l2 {rl2(a, b):.3e}", end="")
        if "cat" in mine:
            c = a.shape[1] // 2
            print(f"   lower(upconv) {rl2(a[:, :c], b[:, :c]):.3e}  upper(skip) {rl2(a[:, c:], b[:, c:]):.3e}")
        else:
            print()
    # at_k grads live in the upper halves; compare with oracle at grads
    for mine, theirs in [("g.cat1", "at3"), ("g.cat2", "at2"), ("g.cat3", "at1")]:
        a = read(mine)
        c = a.shape[1] // 2
        print(f"{mine} upper vs d({theirs}) {rl2(a[:, c:], inter[theirs].grad):.3e}")
    for name, p in net.named_parameters():
        if name.startswith("attns.") and not (".blocks.0.0.layers.0." in name or "out_layer" in name or "embed" in name):
            continue
        g = ref[name].grad
        print(f"  {name:55s} |g|={g.norm():.3e} rel-l2={rl2(p.grad, g):.3e}")


if __name__ == "__main__":
    main()
