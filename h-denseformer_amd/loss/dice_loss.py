"""Drop-in `loss.dice_loss.DiceLoss` (reference loss/dice_loss.py:53-87) on the fused HIP loss kernels: softmax over
classes, per-class soft Dice (BinaryDiceLoss defaults: smooth 1e-5, p 1, batch mean, :5-50) over the classes other than
ignore_index, each times its class weight, divided by C-1 (C when ignore_index is None).  One pass over the logits
(hdf_loss_weighted_forward with the cross-entropy term weighted 0).  Other BinaryDiceLoss settings raise (there is no
eager fallback)."""
from torch import nn

from hdf_rt.loss_fn import DeepSuperCEDice

from .combine_loss import check_dice_kwargs


class DiceLoss(nn.Module):
    def __init__(self, weight=None, ignore_index=None, **kwargs):
        super().__init__()
        self.weight, self.ignore_index, self.kwargs = weight, ignore_index, kwargs

    def forward(self, predict, target):
        assert predict.shape == target.shape, "predict & target shape do not match"
        check_dice_kwargs(self.kwargs)
        return DeepSuperCEDice.apply((target, 0.0, 1.0, self.weight, self.ignore_index), predict)
