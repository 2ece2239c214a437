"""Data-parallel training, one process per GPU, RCCL over xGMI via torch.distributed (backend "nccl" on
ROCm; "gloo" for the CPU tests).  Replaces the reference's single-process nn.DataParallel
(trainer.py:228-229): the batch is sharded across ranks, every rank owns a full replica, and the only
exchange is ONE gradient all-reduce per step over the module's flat fp32 gradient buffer, split in three
buckets that follow the three backward stages (encoder/decoder/heads 26 MB, UpConv chain 19 MB, transformer
branches 17 MB at n_filters=32): a bucket is in flight on a side stream while the next stage back-propagates, so only
the last one is exposed.  On the GPU the default is finer (round 6): ONE backward call that reports FIVE buckets -- decoder
+ heads, encoder levels 1-3, UpConv chain, transformer branches, encoder level 0 -- each at the moment it becomes final;
what is final only with the last kernel of the backward is then 0.1 MB instead of 26.

Loss terms are batch means (dice_loss.py:41, CrossEntropyLoss 'mean'), so with equal per-rank batches
the global gradient is the mean of the rank gradients."""
import torch
import torch.distributed as dist


def flat_allreduce_mean(flat, world, group=None):
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat.div_(world)


def bucket_bounds(table, chain_name="deep_conv.double_conv.0.weight", unet_name="block_1_1_left.conv.weight"):
    """(lo, hi) float ranges of the three gradient buckets in state_dict order:
    [transformer branches | UpConv chain | encoder/decoder/heads]; returned as (stage1, stage2, stage3) ranges."""
    chain = next(off for (name, off, _n, _s) in table if name == chain_name)
    unet = next(off for (name, off, _n, _s) in table if name == unet_name)
    end = max(off + n for (_name, off, n, _s) in table)
    end = (end + 15) // 16 * 16
    return (unet, end), (chain, unet), (0, chain)


class GradSync:
    """Attach to an HDenseFormer: model.grad_hook = GradSync(model).  Call .wait() before optimizer.step().

    Default protocol (GPU): the module runs its backward as ONE call (hdf_backward_events) and hands this hook five
    events, "bucket k is final", recorded inside the call on whichever of its streams finishes the bucket; the hook's
    communication stream waits for each event and all-reduces that bucket (Runtime.grad_buckets) -- in the order the
    buckets become final (decoder + heads, encoder levels 1-3, UpConv chain, transformer branches, encoder level 0) --
    while the rest of the backward is still running.
    staged=True (and every CPU run) keeps the three staged backward calls with one hook call after each."""

    # buckets in the order hdf_backward_events finishes them (index into the event list / Runtime.grad_buckets())
    EVENT_ORDER = (0, 3, 1, 2, 4)

    def __init__(self, model, group=None, staged=False, high_priority=False):
        self.model, self.group = model, group
        self.staged = staged or not torch.cuda.is_available()
        self.world = dist.get_world_size(group)
        # communication stream.  high_priority=True gives it the highest stream priority; measured with stand-in collectives
        # (tools/stage_cost.py, profiles/r06_stage_cost.json) it made no consistent difference for the one-call protocol and
        # cost the staged RCCL world-1 path 5 ms per step (16.3 vs 11.1 ms: the plan's branch stream has that priority too),
        # so the default stays the normal priority of rounds 3-5
        self.comm = None
        if torch.cuda.is_available():
            try:
                lo, hi = torch.cuda.Stream.priority_range()
            except Exception:
                lo, hi = 0, -1
            self.comm = torch.cuda.Stream(priority=hi if high_priority else lo)
        self._pending = []
        flat = model.flat_parameters()
        dist.broadcast(flat, src=0, group=group)          # replicas start identical (trainer seeds after init)
        from . import _lib
        self.buckets = bucket_bounds(model._plan(_lib.F32).table)   # index = stage - 1

    def __call__(self, stage):
        g = self.model.flat_grads()
        lo, hi = self.buckets[stage - 1]
        hi = min(hi, g.numel())
        chunk = g[lo:hi]
        if self.comm is None:
            flat_allreduce_mean(chunk, self.world, self.group)
            return
        ready = torch.cuda.Event()
        ready.record()
        with torch.cuda.stream(self.comm):
            self.comm.wait_event(ready)
            self._reduce(chunk)
            done = torch.cuda.Event()
            done.record()
        self._pending.append(done)

    def _reduce(self, chunk):
        flat_allreduce_mean(chunk, self.world, self.group)

    def on_bucket_events(self, events, ranges):
        """events[k] fires when gradient bucket k = floats ranges[k] = (lo, hi) of the flat buffer is final"""
        from . import _lib
        g = self.model.flat_grads()
        with torch.cuda.stream(self.comm):
            for k in self.EVENT_ORDER:
                lo, hi = ranges[k]
                if hi <= lo:
                    continue
                _lib.check(_lib.lib().hdf_stream_wait_event(self.comm.cuda_stream, events[k]), "hdf_stream_wait_event")
                self._reduce(g[lo:min(hi, g.numel())])
            done = torch.cuda.Event()
            done.record()
        self._pending.append(done)

    def wait(self):
        for ev in self._pending:
            torch.cuda.current_stream().wait_event(ev)
        self._pending = []
