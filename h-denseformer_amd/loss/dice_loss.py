"""Drop-in `loss.dice_loss.DiceLoss` (reference loss/dice_loss.py:53-87) on the fused HIP loss kernels: softmax over
classes, per-class soft Dice (BinaryDiceLoss defaults: smooth 1e-5, p 1, batch mean, :5-50) over classes != 0,
divided by C-1.  One pass over the logits (hdf_loss_terms_forward with the cross-entropy term weighted 0).
Supported: weight=None, ignore_index=0 and the BinaryDiceLoss defaults -- what trainer.py:763-765 builds; anything
else raises (there is no eager fallback)."""
from torch import nn

from hdf_rt.loss_fn import DeepSuperCEDice


class DiceLoss(nn.Module):
    def __init__(self, weight=None, ignore_index=None, **kwargs):
        super().__init__()
        self.weight, self.ignore_index, self.kwargs = weight, ignore_index, kwargs

    def forward(self, predict, target):
        assert predict.shape == target.shape, "predict & target shape do not match"
        if self.weight is not None or self.ignore_index != 0 or self.kwargs:
            raise NotImplementedError("fused DiceLoss supports weight=None, ignore_index=0, default BinaryDiceLoss kwargs")
        return DeepSuperCEDice.apply((target, 0.0, 1.0), predict)
