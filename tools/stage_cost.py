"""What the data-parallel plumbing costs on ONE GPU at the benchmark geometry (4x128^3, batch 2, bf16):
   a) backward as one call (stages=7), no hook                     -- what `python bench.py` times
   b) backward in three stages (1, 2, 4) with a hook that does nothing -- the cost of the three side-stream joins
   c) three stages + GradSync over RCCL with world_size 1          -- + the bucket all-reduces on the comm stream
Run under `python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 ... tools/stage_cost.py`
(or plainly: it then sets up a one-rank rendezvous on 127.0.0.1 itself).  Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "h-denseformer_amd")):
    sys.path.insert(0, p)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
import torch.distributed as dist


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29591")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", device_id=dev)
    from hdf_rt.optim import FlatAdam
    from hdf_rt.parallel import GradSync
    from loss.combine_loss import CEPlusDice, DeepSuperloss
    from models.HDenseFormer import HDenseFormer

    size = int(os.environ.get("HDF_STAGE_COST_SIZE", "128"))
    steps = int(os.environ.get("HDF_STAGE_COST_STEPS", "12"))
    torch.manual_seed(0)
    net = HDenseFormer(4, 4, 32, image_size=(size,) * 3, transformer_depth=24).to(dev)
    net.train()
    net.compute_dtype = "bf16"
    crit = DeepSuperloss(criterion=CEPlusDice(weight=None, ignore_index=0))
    opt = FlatAdam(net, lr=1e-3, weight_decay=1e-4)
    g = torch.Generator(device="cpu").manual_seed(1234)
    x = torch.rand(2, 4, size, size, size, generator=g).to(dev)
    lab = torch.randint(0, 4, (2, size, size, size), generator=g)
    t = torch.nn.functional.one_hot(lab, 4).permute(0, 4, 1, 2, 3).float().contiguous().to(dev)
    sync = GradSync(net)                      # one-call backward + bucket events (default protocol)
    sync_staged = GradSync(net, staged=True)  # round-3 protocol: three staged backward calls

    def step(seed=None):
        if seed is not None:
            net.set_dropout_seed(seed)
        opt.zero_grad()
        loss = crit(net(x), t)
        loss.backward()
        if net.grad_hook is sync or net.grad_hook is sync_staged:
            net.grad_hook.wait()
        return loss

    def timed(hook):
        net.grad_hook = hook
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        evs[0].record()
        for k in range(steps):
            step()
            evs[k + 1].record()
        torch.cuda.synchronize()
        ts = sorted(evs[k].elapsed_time(evs[k + 1]) for k in range(steps))
        return ts[len(ts) // 2]

    # gradients: synced through RCCL (world 1: all-reduce + divide by 1) == the plain backward, same masks
    net.grad_hook = None
    step(seed=77)
    torch.cuda.synchronize()
    g_plain = net.flat_grads().clone()
    net.grad_hook = sync
    step(seed=77)
    torch.cuda.synchronize()
    g_sync = net.flat_grads().clone()
    err = float((g_sync - g_plain).abs().max() / (g_plain.abs().max() + 1e-30))

    rec = {"world": dist.get_world_size(), "backend": dist.get_backend(), "size": size, "grad_rel_err": err,
           "ms_one_call": timed(None), "ms_three_stages_noop_hook": timed(lambda stage: None), "ms_three_stages_rccl": timed(sync_staged), "ms_one_call_events_rccl": timed(sync)}
    print(json.dumps(rec), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
