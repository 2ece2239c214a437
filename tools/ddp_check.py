"""Data-parallel check on ONE GPU: exercises GradSync, the staged backward and the comm-stream overlap logic on the
real HIP path.  Default: two ranks on cuda:0 over gloo.  HDF_DDP_BACKEND=nccl with --nproc-per-node 1: the same
checks through RCCL with one rank (RCCL refuses two ranks on one device): init_process_group("nccl", device_id=..),
the broadcast and the three bucket all-reduces on the comm stream run as RCCL kernels.  Launched by
tests/test_gpu_model.py through torch.distributed.run.  Prints one JSON line per rank."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "h-denseformer_amd")):
    sys.path.insert(0, p)
import torch
import torch.distributed as dist


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    backend = os.environ.get("HDF_DDP_BACKEND", "gloo")
    if backend == "nccl":
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group("gloo")
    from hdf_rt.optim import FlatAdam
    from hdf_rt.parallel import GradSync
    from loss.combine_loss import CEPlusDice, DeepSuperloss
    from models.HDenseFormer import HDenseFormer

    size = (32, 32, 32)
    torch.manual_seed(100 + rank)  # different init per rank: the broadcast in GradSync must equalise them
    net = HDenseFormer(4, 3, 32, image_size=size, transformer_depth=24).to(dev)
    net.train()
    net.compute_dtype = "fp32"
    crit = DeepSuperloss(criterion=CEPlusDice(weight=None, ignore_index=0))
    sync = GradSync(net)
    net.grad_hook = sync
    opt = FlatAdam(net, lr=1e-3, weight_decay=1e-4)

    def data(r):
        g = torch.Generator(device="cpu").manual_seed(7 + r)
        x = torch.rand(1, 4, *size, generator=g).to(dev)
        lab = torch.randint(0, 3, (1, *size), generator=g)
        t = torch.nn.functional.one_hot(lab, 3).permute(0, 4, 1, 2, 3).float().contiguous().to(dev)
        return x, t

    x, t = data(rank)
    net._step = 4  # forward() bumps it to 5: every forward below draws the same dropout masks
    opt.zero_grad()
    loss = crit(net(x), t)
    loss.backward()
    sync.wait()
    torch.cuda.synchronize()
    g_sync = net.flat_grads().clone()

    # reference: the mean of the two ranks' LOCAL gradients, computed without the hook
    net.grad_hook = None
    locals_ = []
    for r in range(world):
        xr, tr = data(r)
        net._step = 4
        net.set_dropout_seed(net.step_seed(5, r))   # rank r's masks (the seed folds the rank in)
        opt.zero_grad()
        crit(net(xr), tr).backward()
        torch.cuda.synchronize()
        locals_.append(net.flat_grads().clone())
    g_ref = sum(locals_) / world
    err = float((g_sync - g_ref).abs().max() / (g_ref.abs().max() + 1e-30))

    # parameters identical across ranks after the broadcast + one optimizer step on synced gradients
    net.grad_hook = sync
    net._step = 4
    opt.zero_grad()
    crit(net(x), t).backward()
    sync.wait()
    opt.step()
    torch.cuda.synchronize()
    flat = net.flat_parameters().detach()
    if backend != "nccl":
        flat = flat.cpu()
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    pdiff = float((gathered[0] - gathered[-1]).abs().max())
    print(json.dumps({"rank": rank, "world": world, "backend": dist.get_backend(), "grad_rel_err": err,
                      "param_max_diff": pdiff, "loss": float(loss.item())}), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
