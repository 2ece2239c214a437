"""Data-parallel training, one process per GPU, RCCL over xGMI via torch.distributed (backend "nccl" on
ROCm; "gloo" for the CPU tests).  Replaces the reference's single-process nn.DataParallel
(trainer.py:228-229): the batch is sharded across ranks, every rank owns a full replica, and the only
exchange is ONE gradient all-reduce per step over the module's flat fp32 gradient buffer, split in two
buckets that follow the two backward stages so the first (encoder/decoder/heads, ~2/3 of the bytes) is
in flight on a side stream while the transformer branches are still back-propagating.

Loss terms are batch means (dice_loss.py:41, CrossEntropyLoss 'mean'), so with equal per-rank batches
the global gradient is the mean of the rank gradients."""
import torch
import torch.distributed as dist


def flat_allreduce_mean(flat, world, group=None):
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat.div_(world)


def bucket_bounds(table, split_name="block_1_1_left.conv.weight"):
    """(lo, hi) float ranges of the two gradient buckets: [stage-2 params | stage-1 params]."""
    split = next(off for (name, off, _n, _s) in table if name == split_name)
    end = max(off + n for (_name, off, n, _s) in table)
    end = (end + 15) // 16 * 16
    return (0, split), (split, end)


class GradSync:
    """Attach to an HDenseFormer: model.grad_hook = GradSync(model).  Call .wait() before optimizer.step()."""

    def __init__(self, model, group=None):
        self.model, self.group = model, group
        self.world = dist.get_world_size(group)
        self.comm = torch.cuda.Stream() if torch.cuda.is_available() else None
        self._pending = []
        flat = model.flat_parameters()
        dist.broadcast(flat, src=0, group=group)          # replicas start identical (trainer seeds after init)
        from . import _lib
        self.b2, self.b1 = bucket_bounds(model._plan(_lib.F32).table)

    def __call__(self, stage):
        g = self.model.flat_grads()
        lo, hi = self.b1 if stage == 1 else self.b2
        hi = min(hi, g.numel())
        chunk = g[lo:hi]
        if self.comm is None:
            flat_allreduce_mean(chunk, self.world, self.group)
            return
        ready = torch.cuda.Event()
        ready.record()
        with torch.cuda.stream(self.comm):
            self.comm.wait_event(ready)
            flat_allreduce_mean(chunk, self.world, self.group)
            done = torch.cuda.Event()
            done.record()
        self._pending.append(done)

    def wait(self):
        for ev in self._pending:
            torch.cuda.current_stream().wait_event(ev)
        self._pending = []
