"""Drop-in `loss.combine_loss` (reference loss/combine_loss.py:8-35,68-79) backed by the fused HIP loss.

Supported: what trainer.py:743-771 (_get_loss) builds for the H-DenseFormer runs --
DeepSuperloss(criterion=CEPlusDice(weight=class_weight or None, ignore_index=0)) -- plus ignore_index=None, with the
BinaryDiceLoss defaults (smooth 1e-5, p 1, reduction 'mean').  Anything else raises: there is no eager fallback."""
from torch import nn

from hdf_rt.loss_fn import DeepSuperCEDice

_DICE_DEFAULTS = dict(smooth=1e-5, p=1, reduction="mean")


def check_dice_kwargs(kwargs):
    """BinaryDiceLoss keyword arguments (dice_loss.py:20-25): the kernels implement the defaults (trainer.py:761 passes
    p=1 explicitly); k only matters for reduction='topk'."""
    for k, v in kwargs.items():
        if k == "k":
            continue
        if k not in _DICE_DEFAULTS or v != _DICE_DEFAULTS[k]:
            raise NotImplementedError(f"fused Dice loss implements BinaryDiceLoss(smooth=1e-5, p=1, reduction='mean'); "
                                      f"got {k}={v!r}")


class CEPlusDice(nn.Module):
    def __init__(self, weight=None, ignore_index=None, **kwargs):
        super().__init__()
        self.weight, self.ignore_index, self.kwargs = weight, ignore_index, kwargs

    def _check(self):
        check_dice_kwargs(self.kwargs)

    def _spec(self, target):
        return (target, 1.0, 1.0, self.weight, self.ignore_index)

    def forward(self, predict, target):
        assert predict.size() == target.size()
        self._check()
        return DeepSuperCEDice.apply(self._spec(target), predict)


class DeepSuperloss(nn.Module):
    def __init__(self, criterion=None):
        super().__init__()
        self.loss = criterion

    def forward(self, input, target):
        if not isinstance(self.loss, CEPlusDice):
            raise NotImplementedError("fused DeepSuperloss needs criterion=CEPlusDice(...)")
        self.loss._check()
        return DeepSuperCEDice.apply(self.loss._spec(target), *list(input))
