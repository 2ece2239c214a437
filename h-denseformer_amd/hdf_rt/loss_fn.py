"""Fused DeepSuperloss(CEPlusDice) on the GPU: one autograd node, one pass over the logits per scale
(hdf_loss_forward / hdf_loss_backward in include/hdf.h).  Reference: loss/combine_loss.py:8-35,68-79."""
import torch

from . import _lib
from ._lib import BF16, F16, F32, check, lib, ptr, stream_ptr

_DT = {torch.float32: F32, torch.bfloat16: BF16, torch.float16: F16}

# class weights as the trainer builds them (trainer.py:743-771: a CPU tensor / list) -> one device copy per (device,
# values): a pageable host-to-device copy on every loss call would block the host once per step
_CW_CACHE = {}


def _device_class_weight(cw, device):
    if torch.is_tensor(cw) and cw.is_cuda:
        return cw.detach().to(device=device, dtype=torch.float32).contiguous()
    vals = tuple(float(v) for v in (cw.detach().flatten().tolist() if torch.is_tensor(cw) else list(cw)))
    shape = tuple(cw.shape) if torch.is_tensor(cw) else (len(vals),)
    key = (str(device), shape, vals)
    t = _CW_CACHE.get(key)
    if t is None:
        if len(_CW_CACHE) > 64:
            _CW_CACHE.clear()
        t = _CW_CACHE[key] = torch.tensor(vals, dtype=torch.float32).reshape(shape).to(device)
    return t


class DeepSuperCEDice(torch.autograd.Function):
    """apply(target, *outs) = DeepSuperloss(CEPlusDice(weight=None, ignore_index=0)); apply((target, w_ce, w_dice), *outs)
    weights the two terms (0,1: DiceLoss(ignore_index=0); 1,0: CrossentropyLoss); apply((target, w_ce, w_dice,
    class_weight, ignore_index), *outs) adds the per-class weights and the Dice ignore_index (None: all classes) that
    trainer.py:743-771 can pass."""

    @staticmethod
    def forward(ctx, target, *outs):
        w_ce, w_dice, cw, ignore = 1.0, 1.0, None, 0
        if isinstance(target, tuple):
            if len(target) == 3:
                target, w_ce, w_dice = target
            else:
                target, w_ce, w_dice, cw, ignore = target
        if not all(o.is_cuda for o in outs) or not target.is_cuda:
            raise _lib.HdfError("fused loss needs GPU tensors (no CPU fallback)")
        dt = outs[0].dtype
        if dt not in _DT or any(o.dtype != dt for o in outs):
            raise _lib.HdfError(f"fused loss: logits must all be float32, all bfloat16 or all float16 (got {[o.dtype for o in outs]})")
        n = len(outs)
        if not 1 <= n <= 4:
            raise _lib.HdfError("fused loss handles 1..4 deep-supervision scales")
        if target.dim() not in (4, 5):
            raise _lib.HdfError(f"fused loss: target must be [B,C,D,H,W] or [B,C,H,W], got {tuple(target.shape)}")
        sp = tuple(target.shape[2:])
        b, c = target.shape[:2]
        d, h, w = ((1,) + sp) if len(sp) == 2 else sp        # 2-D logits (HDenseFormer_2D): depth 1
        for i, o in enumerate(outs):
            if tuple(o.shape) != (b, c) + tuple(v >> i for v in sp):
                raise AssertionError(f"predict & target shape do not match at scale {i}: {tuple(o.shape)}")
        outs = [o.contiguous() for o in outs]
        tgt = target.float().contiguous()
        if cw is not None:
            cw = _device_class_weight(cw, tgt.device)
            if cw.dim() != 1 or cw.shape[0] != c:
                raise AssertionError(f"Expect weight shape [{c}], get[{tuple(cw.shape)}]")     # dice_loss.py:80-81
        ign = -1 if ignore is None else int(ignore)
        ws = torch.empty(lib().hdf_loss_workspace_bytes(b), dtype=torch.uint8, device=tgt.device)
        loss = torch.empty((), dtype=torch.float32, device=tgt.device)
        po = [ptr(o) for o in outs] + [None] * (4 - n)
        check(lib().hdf_loss_weighted_forward(_DT[dt], po[0], po[1], po[2], po[3], n, ptr(tgt), b, c, d, h, w,
                                              float(w_ce), float(w_dice), ptr(cw), ign, ptr(ws), ptr(loss),
                                              stream_ptr()), "hdf_loss_weighted_forward")
        ctx.save_for_backward(tgt, ws, *outs)
        ctx.n, ctx.w, ctx.dhw, ctx.cw, ctx.ign = n, (float(w_ce), float(w_dice)), (d, h, w), cw, ign
        return loss

    @staticmethod
    def backward(ctx, g):
        tgt, ws, *outs = ctx.saved_tensors
        n = ctx.n
        b, c = tgt.shape[:2]
        d, h, w = ctx.dhw
        douts = [torch.empty_like(o) for o in outs]
        gg = g.detach().float().reshape(1).contiguous()
        po = [ptr(o) for o in outs] + [None] * (4 - n)
        pd = [ptr(o) for o in douts] + [None] * (4 - n)
        check(lib().hdf_loss_weighted_backward(_DT[outs[0].dtype], po[0], po[1], po[2], po[3], n, ptr(tgt), b, c, d, h,
                                               w, ctx.w[0], ctx.w[1], ptr(ctx.cw), ctx.ign, ptr(ws), ptr(gg), pd[0],
                                               pd[1], pd[2], pd[3], stream_ptr()), "hdf_loss_weighted_backward")
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
    return np.nanmean(vals[1:])          # numpy scalar like the reference's (its call site does dice.item())


class RunningDice:
    """Drop-in for metrics.RunningDice (metrics.py:82-151) with the confusion matrix accumulated on the GPU.

    update_matrix(ground_truth, prediction) takes exactly what the reference takes -- two CLASS MAPS (numpy arrays or
    tensors of any integer / float dtype and any shape; trainer.py:393-398 passes the argmax maps).  The pair
    (one-hot target, logits) straight from the step goes through update_from_logits(), which never materialises an
    argmax map; the two forms are separate methods because a float class map of shape [B, H, W] cannot be told from
    a score tensor by dtype and shape.  As in the reference, an update whose ground truth consists only of
    `ignore_label` is dropped (metrics.py:122-124) -- decided on the device, without a host sync.
    compute_dice() does one 512-byte D2H."""

    def __init__(self, labels, ignore_label=0):
        self.labels = list(labels)
        self.ignore_label = ignore_label
        self.conf = None

    def _accumulate(self, cur, all_ignore):
        # `if (ground_truth == self.ignore_label).all(): return`: the update is multiplied by 0 instead
        cur = cur * (~all_ignore).to(cur.dtype)
        self.conf = cur if self.conf is None else self.conf + cur

    def update_from_logits(self, target_onehot, logits):
        """target_onehot, logits: [B, C, *spatial] floating tensors on the GPU with C == len(labels)."""
        c = len(self.labels)
        if not (torch.is_tensor(target_onehot) and torch.is_tensor(logits) and logits.is_cuda and logits.dim() >= 3
                and target_onehot.shape == logits.shape and logits.shape[1] == c):
            raise _lib.HdfError(f"RunningDice.update_from_logits: expected two [B, {c}, ...] GPU tensors of one shape, "
                                f"got {getattr(target_onehot, 'shape', None)} and {getattr(logits, 'shape', None)}")
        lg = logits.detach().contiguous()
        tg = target_onehot.detach().float().contiguous()
        if lg.dtype not in _DT:
            raise _lib.HdfError(f"RunningDice: unsupported logits dtype {lg.dtype}")
        cur = torch.zeros((8, 8), dtype=torch.int64, device=lg.device)
        check(lib().hdf_confusion_matrix(_DT[lg.dtype], ptr(lg), ptr(tg), lg.shape[0], lg.shape[1],
                                         lg[0, 0].numel(), ptr(cur), 0, stream_ptr()), "hdf_confusion_matrix")
        ig = self.ignore_label
        rows = cur.sum(1)       # every voxel is counted (argmax < C): all-ignore <=> only row `ig` is populated
        all_ignore = (rows.sum() == rows[ig]) if isinstance(ig, int) and 0 <= ig < c else torch.zeros((), dtype=torch.bool, device=lg.device)
        self._accumulate(cur, all_ignore)

    def update_matrix(self, ground_truth, prediction):
        c = len(self.labels)
        dev = self.conf.device if self.conf is not None else (
            prediction.device if torch.is_tensor(prediction) and prediction.is_cuda else
            ground_truth.device if torch.is_tensor(ground_truth) and ground_truth.is_cuda else
            torch.device("cuda", torch.cuda.current_device()))
        if dev.type != "cuda":
            raise _lib.HdfError("RunningDice needs a GPU (there is no CPU path)")
        gt0 = torch.as_tensor(ground_truth).to(dev).flatten()
        pr = torch.as_tensor(prediction).to(dev).flatten()
        if gt0.numel() != pr.numel():
            raise ValueError("ground_truth and prediction differ in size")
        all_ignore = (gt0 == self.ignore_label).all()        # over EVERY voxel, like the reference (labels >= n_cls too)
        # labels outside [0, 255) (e.g. a negative ignore value) must not alias a class after the uint8 cast
        gt = torch.where((gt0 >= 0) & (gt0 < 255), gt0, torch.full_like(gt0, 255)).to(torch.uint8).contiguous()
        pr = torch.where((pr >= 0) & (pr < 255), pr, torch.full_like(pr, 255)).to(torch.uint8).contiguous()
        cur = torch.zeros((8, 8), dtype=torch.int64, device=dev)
        check(lib().hdf_confusion_matrix_labels(ptr(gt), ptr(pr), c, gt.numel(), ptr(cur), 0, stream_ptr()),
              "hdf_confusion_matrix_labels")
        self._accumulate(cur, all_ignore)

    def compute_dice(self, smooth=1e-5):
        import numpy as np
        c = len(self.labels)
        m = self.conf.cpu().numpy()[:c, :c]
        inter = np.diag(m)
        union = m.sum(axis=1) + m.sum(axis=0)
        iou = (2 * inter + smooth) / (union.astype(np.float32) + smooth)
        return np.mean(iou[1:]), [round(v, 4) for v in iou]

    def init_op(self):
        self.conf = None
