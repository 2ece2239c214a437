"""Drop-in `models.HDenseFormer_2D` for MI355X (reference models/HDenseFormer_2D.py:172-256): same constructor,
factories and state_dict (Conv2d / ConvTranspose2d / InstanceNorm2d parameter shapes), 4-output forward on
[B, C, H, W] inputs, all arithmetic in libhdf_hip.so.

The library runs the 2-D model as its exact depth-replicated 3-D embedding (hdf_plan_create_2d, csrc/plan.hip
"2-D embedding"): every 2-D kernel is placed on the depth taps of a 3-D kernel such that all activations consist of
identical depth slices, the 2-D logits are depth slice 0, and the 2-D parameter gradients are the embedded sums of the
3-D ones.  Same kernels, same parity, at the cost of the redundant slices."""
from .HDenseFormer import HDenseFormer

__all__ = ["HDenseFormer_2D", "HDenseFormer_2D_32", "HDenseFormer_2D_16"]


class HDenseFormer_2D(HDenseFormer):
    _ND = 2

    def __init__(self, in_channels, n_cls, n_filters, image_size=(384, 384), transformer_depth=12):
        super().__init__(in_channels, n_cls, n_filters, image_size=image_size, transformer_depth=transformer_depth)


def HDenseFormer_2D_32(in_channels, n_cls, image_size, transformer_depth):
    return HDenseFormer_2D(in_channels=in_channels, n_cls=n_cls, image_size=image_size, n_filters=32,
                           transformer_depth=transformer_depth)


def HDenseFormer_2D_16(in_channels, n_cls, image_size, transformer_depth):
    return HDenseFormer_2D(in_channels=in_channels, n_cls=n_cls, image_size=image_size, n_filters=16,
                           transformer_depth=transformer_depth)
