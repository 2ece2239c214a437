"""Fused DeepSuperloss(CEPlusDice) on the GPU: one autograd node, one pass over the logits per scale
(hdf_loss_forward / hdf_loss_backward in include/hdf.h).  Reference: loss/combine_loss.py:8-35,68-79."""
import torch

from . import _lib
from ._lib import BF16, F32, check, lib, ptr, stream_ptr

_DT = {torch.float32: F32, torch.bfloat16: BF16}


class DeepSuperCEDice(torch.autograd.Function):
    @staticmethod
    def forward(ctx, target, *outs):
        if not all(o.is_cuda for o in outs) or not target.is_cuda:
            raise _lib.HdfError("fused loss needs GPU tensors (no CPU fallback)")
        dt = outs[0].dtype
        if dt not in _DT or any(o.dtype != dt for o in outs):
            raise _lib.HdfError(f"fused loss: logits must all be float32 or all bfloat16 (got {[o.dtype for o in outs]})")
        n = len(outs)
        if not 1 <= n <= 4:
            raise _lib.HdfError("fused loss handles 1..4 deep-supervision scales")
        b, c, d, h, w = target.shape
        for i, o in enumerate(outs):
            if tuple(o.shape) != (b, c, d >> i, h >> i, w >> i):
                raise AssertionError(f"predict & target shape do not match at scale {i}: {tuple(o.shape)}")
        outs = [o.contiguous() for o in outs]
        tgt = target.float().contiguous()
        ws = torch.empty(lib().hdf_loss_workspace_bytes(b), dtype=torch.uint8, device=tgt.device)
        loss = torch.empty((), dtype=torch.float32, device=tgt.device)
        po = [ptr(o) for o in outs] + [None] * (4 - n)
        check(lib().hdf_loss_forward(_DT[dt], po[0], po[1], po[2], po[3], n, ptr(tgt), b, c, d, h, w, ptr(ws),
                                     ptr(loss), stream_ptr()), "hdf_loss_forward")
        ctx.save_for_backward(tgt, ws, *outs)
        ctx.n = n
        return loss

    @staticmethod
    def backward(ctx, g):
        tgt, ws, *outs = ctx.saved_tensors
        n = ctx.n
        b, c, d, h, w = tgt.shape
        douts = [torch.empty_like(o) for o in outs]
        gg = g.detach().float().reshape(1).contiguous()
        po = [ptr(o) for o in outs] + [None] * (4 - n)
        pd = [ptr(o) for o in douts] + [None] * (4 - n)
        check(lib().hdf_loss_backward(_DT[outs[0].dtype], po[0], po[1], po[2], po[3], n, ptr(tgt), b, c, d, h, w,
                                      ptr(ws), ptr(gg), pd[0], pd[1], pd[2], pd[3], stream_ptr()), "hdf_loss_backward")
        return (None, *douts)


def dice_counts(logits, target_onehot):
    """[B, 8, 3] int64 counts (|P&T|, |P|, |T|) per class from hard argmax (trainer.py:919-945)."""
    lg = logits.detach().contiguous()
    tg = target_onehot.detach().float().contiguous()
    b, c = lg.shape[:2]
    vox = lg[0, 0].numel()
    counts = torch.empty((b, 8, 3), dtype=torch.int64, device=lg.device)
    check(lib().hdf_dice_counts(_DT[lg.dtype], ptr(lg), ptr(tg), b, c, vox, ptr(counts), stream_ptr()),
          "hdf_dice_counts")
    return counts


def compute_dice(logits, target_onehot, ignore_index=0):
    """trainer.compute_dice (trainer.py:919-945) from the on-device counts: one tiny D2H instead of the
    reference's per-class .item() syncs; same rounding (4 dp per class) and absent-class rule."""
    import numpy as np
    cnt = dice_counts(logits, target_onehot).cpu().numpy().astype(np.float64)   # [B,8,3]
    c = logits.shape[1]
    vals = np.ones(c, dtype=np.float32)
    for k in range(c):
        if k == ignore_index:
            continue
        if cnt[:, k, 1].sum() == 0 and cnt[:, k, 2].sum() == 0:
            continue
        d = np.mean((2 * cnt[:, k, 0].astype(np.float32) + np.float32(1e-5)) /
                    (cnt[:, k, 1].astype(np.float32) + cnt[:, k, 2].astype(np.float32) + np.float32(1e-5)))
        vals[k] = round(float(d), 4)
    return float(np.nanmean(vals[1:]))


class RunningDice:
    """Drop-in for metrics.RunningDice (metrics.py:82-151) with the confusion matrix accumulated on the GPU
    (hdf_confusion_matrix): update_matrix takes the logits and the one-hot target directly -- no argmax maps are
    copied to the host -- and compute_dice() does one 512-byte D2H."""

    def __init__(self, labels, ignore_label=0):
        self.labels = list(labels)
        self.ignore_label = ignore_label
        self.conf = None

    def update_matrix(self, target_onehot, logits):
        lg = logits.detach().contiguous()
        tg = target_onehot.detach().float().contiguous()
        b, c = lg.shape[:2]
        first = self.conf is None
        if first:
            self.conf = torch.zeros((8, 8), dtype=torch.int64, device=lg.device)
        check(lib().hdf_confusion_matrix(_DT[lg.dtype], ptr(lg), ptr(tg), b, c, lg[0, 0].numel(), ptr(self.conf),
                                         0 if first else 1, stream_ptr()), "hdf_confusion_matrix")

    def compute_dice(self, smooth=1e-5):
        import numpy as np
        c = len(self.labels)
        m = self.conf.cpu().numpy()[:c, :c]
        inter = np.diag(m)
        union = m.sum(axis=1) + m.sum(axis=0)
        iou = (2 * inter + smooth) / (union.astype(np.float32) + smooth)
        return float(np.mean(iou[1:])), [round(float(v), 4) for v in iou]

    def init_op(self):
        self.conf = None
