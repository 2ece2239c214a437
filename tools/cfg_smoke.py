import os, sys, time
ROOT = "/root/repo"
for p in (ROOT, os.path.join(ROOT, "h-denseformer_amd")):
    sys.path.insert(0, p)
import torch
from hdf_rt.optim import FlatAdam
from loss.combine_loss import CEPlusDice, DeepSuperloss
from models.HDenseFormer import HDenseFormer
dev = torch.device("cuda", 0)
for (cin, ncls, nf, size, td, B) in [(2, 3, 32, 144, 24, 2), (4, 4, 48, 160, 24, 2), (4, 4, 16, 128, 8, 1)]:
    torch.manual_seed(0)
    net = HDenseFormer(cin, ncls, nf, image_size=(size,) * 3, transformer_depth=td).to(dev)
    net.train(); net.compute_dtype = "bf16"
    crit = DeepSuperloss(criterion=CEPlusDice(weight=None, ignore_index=0))
    opt = FlatAdam(net, lr=1e-3, weight_decay=1e-4)
    x = torch.rand(B, cin, size, size, size, device=dev)
    lab = torch.randint(0, ncls, (B, size, size, size))
    t = torch.nn.functional.one_hot(lab, ncls).permute(0, 4, 1, 2, 3).float().contiguous().to(dev)
    losses = []
    for i in range(4):
        if i == 2:
            torch.cuda.synchronize(); t0 = time.time()
        opt.zero_grad(); loss = crit(net(x), t); loss.backward(); opt.step()
        losses.append(float(loss.item()))
    torch.cuda.synchronize(); dt = (time.time() - t0) / 2
    g = net.flat_grads()
    print(f"cfg in={cin} ncls={ncls} nf={nf} {size}^3 td={td} B={B}: losses {losses} finite_grads={bool(torch.isfinite(g).all())} {dt*1e3:.1f} ms/step {B/dt:.1f} samples/s", flush=True)
    del net, opt, x, t
    torch.cuda.empty_cache()
