"""Drop-in `models.HDenseFormer_2D` for MI355X (reference models/HDenseFormer_2D.py:172-256): same constructor,
factories and state_dict (Conv2d / ConvTranspose2d / InstanceNorm2d parameter shapes), 4-output forward on
[B, C, H, W] inputs, all arithmetic in libhdf_hip.so.

Round 6: the library runs the 2-D model natively on depth-1 tensors (hdf_plan_create_2d: 2-D convolutions, transposed
convolutions and weight gradients on the 9 centre-plane taps, MaxPool2d, bilinear x2; csrc/plan.hip "2-D embedding" for
how the 2-D parameters sit in the 27-tap panels).  `net._embedded_2d = True` before the first forward selects the exact
depth-16 replicated 3-D embedding of rounds 3-5 instead (16x ... 2x the arithmetic): the oracle of
tests/test_gpu_model_2d.py."""
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
