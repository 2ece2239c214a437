import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "h-denseformer_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from oracle import detgen, hdf_oracle as orc
from models.HDenseFormer import HDenseFormer
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg = (2, 3, nf, (32, 32, 32), 8)
sd = orc.det_model(*cfg)
net = HDenseFormer(cfg[0], cfg[1], cfg[2], image_size=cfg[3], transformer_depth=cfg[4])
net.load_state_dict(sd); net = net.cuda().eval()
x = torch.from_numpy(detgen.det_input(1, 2, cfg[3], tag="nf64"))
with torch.no_grad():
    outs = net(x.cuda())
    ref, inter = orc.forward(x, sd, None, want_intermediates=True)
rt = net._last_rt
for k, v in inter.items():
    if k.startswith(("dec", "cat")) or k == "at3":
        continue
    g = rt.read_buffer(k)
    print(k, float((g.cpu() - v).abs().max() / v.abs().max()))
for i in range(4):
    print("out", i, float((outs[i].cpu() - ref[i]).abs().max() / ref[i].abs().max()))
