"""What the data-parallel plumbing costs on ONE GPU at the benchmark geometry (4x128^3, batch 2, bf16):
   a) backward as one call (stages=7), no hook                     -- what `python bench.py` times
   b) backward in three stages (1, 2, 4) with a hook that does nothing -- the cost of the three side-stream joins
   c) three stages + GradSync over RCCL with world_size 1          -- + the bucket all-reduces on the comm stream
   d) one call + bucket events + a STAND-IN collective per bucket that holds 16 / 32 compute units for 300 us per 26 MB
      (hdf_op_occupy: 160 KiB of LDS per workgroup, i.e. a compute unit of its own -- what an 8-GPU ring all-reduce of a
      19-26 MB bucket costs over xGMI, on the comm stream, behind the bucket's event)  -- round 6, VERDICT r05 #3: the
      persistent transformer backward needs all 256 compute units resident and is launched while the stand-ins of buckets
      2 and 1 may still hold theirs; the leg reports what that waiting costs and that no barrier gives up
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
    import ctypes as C
    from hdf_rt._lib import check, lib

    class OccupySync(GradSync):
        """GradSync whose collective is a kernel that holds `wgs` compute units for `usec` (no data moved)"""

        def __init__(self, model, wgs, usec, lds, vgprs, high_priority=False):
            super().__init__(model, high_priority=high_priority)
            self.wgs, self.usec, self.lds, self.vgprs = wgs, usec, lds, vgprs

        def _reduce(self, chunk):
            # `usec` is what the largest bucket of the three-bucket protocol (26 MB) costs; a bucket pays by its size
            us = max(10, int(self.usec * chunk.numel() * 4 / 26.0e6))
            check(lib().hdf_op_occupy(self.wgs, self.lds, self.vgprs, us, torch.cuda.current_stream().cuda_stream), "occupy")

    sync = GradSync(net)                      # one-call backward + bucket events (default protocol)
    sync_staged = GradSync(net, staged=True)  # round-3 protocol: three staged backward calls

    def step(seed=None):
        if seed is not None:
            net.set_dropout_seed(seed)
        opt.zero_grad()
        loss = crit(net(x), t)
        loss.backward()
        if isinstance(net.grad_hook, GradSync):
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

    rec = {"world": dist.get_world_size(), "backend": dist.get_backend(), "size": size, "grad_rel_err": err}
    # Legs, measured in ROUNDS (every leg once per round, in this order, so that drift of the box shows up as a difference
    # between rounds rather than between legs).  Stand-in collectives: "rccl_like" = 128 VGPRs + 16 KiB of LDS per workgroup;
    # "whole_cu" = 160 KiB of LDS (shares a unit with nothing).  "budget224": the persistent kernels' grids sized for 224 of
    # the 256 units (32 left to the collective: hdf_set_cu_budget; the transformer branches then run as the launch chain).
    legs = [("ms_one_call", 256, lambda: None),
            ("ms_three_stages_noop_hook", 256, lambda: (lambda stage: None)),
            ("ms_three_stages_rccl", 256, lambda: sync_staged),
            ("ms_one_call_events_rccl", 256, lambda: sync)]
    for tag, lds, vg in (("rccl_like", 16 * 1024, 128), ("whole_cu", 160 * 1024, 0)):
        for wgs in (16, 32):
            legs.append(("ms_one_call_events_standin_%s_%dwg_300us" % (tag, wgs), 256,
                         lambda wgs=wgs, lds=lds, vg=vg: OccupySync(net, wgs, 300, lds, vg)))
    legs.append(("ms_one_call_events_standin_whole_cu_32wg_300us_high_priority_comm_stream", 256,
                 lambda: OccupySync(net, 32, 300, 160 * 1024, 0, high_priority=True)))
    legs.append(("ms_one_call_budget224", 224, lambda: None))
    for tag, lds, vg in (("rccl_like", 16 * 1024, 128), ("whole_cu", 160 * 1024, 0)):
        legs.append(("ms_one_call_events_budget224_standin_%s_32wg_300us" % tag, 224,
                     lambda lds=lds, vg=vg: OccupySync(net, 32, 300, lds, vg)))
    rounds = int(os.environ.get("HDF_STAGE_COST_ROUNDS", "2"))
    for name, _b, _m in legs:
        rec[name] = []
    for _ in range(rounds):
        for name, budget, make in legs:
            check(lib().hdf_set_cu_budget(budget), "budget")
            rec[name].append(round(timed(make()), 4))
    check(lib().hdf_set_cu_budget(256), "budget")
    pers, who = C.c_int(), C.c_int()
    check(lib().hdf_plan_chain_state(net._last_rt.plan.h, 2, C.byref(pers), C.byref(who)), "chain_state")
    rec["persistent_kernels_still_on"] = pers.value
    rec["gave_up_workgroup"] = who.value
    print(json.dumps(rec), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
