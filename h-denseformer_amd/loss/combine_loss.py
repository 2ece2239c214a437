"""Drop-in `loss.combine_loss` (reference loss/combine_loss.py:8-35,68-79) backed by the fused HIP loss.

Supported configuration = the one the trainer builds for the H-DenseFormer runs (trainer.py:224-226,
763-765): DeepSuperloss(criterion=CEPlusDice(weight=None, ignore_index=0)) with BinaryDiceLoss defaults
(smooth 1e-5, p 1, reduction 'mean').  Anything else raises -- there is no eager fallback."""
from torch import nn

from hdf_rt.loss_fn import DeepSuperCEDice


class CEPlusDice(nn.Module):
    def __init__(self, weight=None, ignore_index=None, **kwargs):
        super().__init__()
        self.weight, self.ignore_index, self.kwargs = weight, ignore_index, kwargs

    def _check(self):
        if self.weight is not None or self.ignore_index != 0 or self.kwargs:
            raise NotImplementedError("fused CEPlusDice supports weight=None, ignore_index=0, default Dice kwargs "
                                      "(the configuration trainer.py:763-765 uses)")

    def forward(self, predict, target):
        assert predict.size() == target.size()
        self._check()
        return DeepSuperCEDice.apply(target, predict)


class DeepSuperloss(nn.Module):
    def __init__(self, criterion=None):
        super().__init__()
        self.loss = criterion

    def forward(self, input, target):
        if not isinstance(self.loss, CEPlusDice):
            raise NotImplementedError("fused DeepSuperloss needs criterion=CEPlusDice(weight=None, ignore_index=0)")
        self.loss._check()
        return DeepSuperCEDice.apply(target, *list(input))
