"""One bf16 training step at the g4 geometry (4x64^3, n_filters 32, depth 8, batch 2, dropout seed 4321) through the
drop-in surface; prints ONE JSON line with the loss, per-tensor gradient norms and a checksum of every convolution
weight gradient (those are summed in a fixed order: any stream arrangement must reproduce them bit for bit).
tests/test_gpu_knobs.py runs it once per surviving environment knob (they are read once per process)."""
import json
import os
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "h-denseformer_amd")):
    sys.path.insert(0, p)
import torch

from oracle import detgen
from oracle import hdf_oracle as orc


def main():
    from loss.combine_loss import CEPlusDice, DeepSuperloss
    from models.HDenseFormer import HDenseFormer
    cfg, batch, tag = (4, 4, 32, (64, 64, 64), 8), 2, "g4_mid_train"
    net = HDenseFormer(cfg[0], cfg[1], cfg[2], image_size=cfg[3], transformer_depth=cfg[4])
    net.load_state_dict(orc.det_model(*cfg))
    net = net.to("cuda:0")
    net.compute_dtype = "bf16"
    x = torch.from_numpy(detgen.det_input(batch, cfg[0], cfg[3], tag=tag)).cuda()
    onehot = torch.from_numpy(detgen.one_hot(detgen.det_labels(batch, cfg[1], cfg[3], tag=tag), cfg[1])).cuda()
    crit = DeepSuperloss(criterion=CEPlusDice(weight=None, ignore_index=0))
    rec = {}
    for rep in range(2):          # twice: the second step runs with every lazily created stream / event in place
        net.train()
        net.set_dropout_seed(4321)
        for p in net.parameters():
            p.grad = None
        loss = crit(net(x), onehot)
        loss.backward()
        torch.cuda.synchronize()
        norms, crcs = {}, {}
        for n, p in net.named_parameters():
            g = p.grad.detach()
            norms[n] = float(g.double().norm())
            if n.endswith("conv.weight") or n.endswith("double_conv.0.weight") or (n.startswith("upconv_") and n.endswith(".weight")):
                crcs[n] = zlib.crc32(g.cpu().numpy().tobytes())
        rec[f"step{rep}"] = {"loss": float(loss.item()), "norms": norms, "crcs": crcs}
    print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
