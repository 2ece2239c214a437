"""`loss.dice_loss` names of the reference (loss/dice_loss.py:53-87).  On the MI355X build the Dice term
only exists fused with the cross-entropy (see loss/combine_loss.py); a stand-alone DiceLoss is outside
the hot path and raises."""
from torch import nn


class DiceLoss(nn.Module):
    def __init__(self, weight=None, ignore_index=None, **kwargs):
        super().__init__()
        self.class_weight, self.ignore_index, self.kwargs = weight, ignore_index, kwargs

    def forward(self, predict, target):
        raise NotImplementedError("stand-alone DiceLoss is not part of the MI355X hot path; use "
                                  "CEPlusDice / DeepSuperloss(CEPlusDice) from loss.combine_loss")
