"""Sliding-window inference on the device (reference: SemanticSeg.inference_slidingwindow / cal_steps,
trainer.py:488-618).  The reference runs every window through the net, takes the softmax on the GPU, then adds into two
full-size fp32 accumulators with indexed torch ops and finally argmaxes a softmax of their quotient on the GPU and
moves it to the host.  Here the per-window tail (softmax + accumulate + count) is ONE HIP kernel reading the logits
once, the vote is one kernel writing a uint8 map; the model is the MI355X-native HDenseFormer.  No CPU fallback."""
import numpy as np
import torch

from . import _lib
from ._lib import check, lib, ptr, stream_ptr


def cal_steps(image_size, patch_size, step_size):
    """Window origins per axis, exactly trainer.py:595-618 (the last window ends at the volume's end)."""
    steps = []
    for dim in range(len(image_size)):
        if image_size[dim] <= patch_size[dim]:
            steps.append([0])
            continue
        max_step_value = image_size[dim] - patch_size[dim]
        num_steps = int(np.ceil(max_step_value / step_size[dim])) + 1
        actual_step_size = max_step_value / (num_steps - 1)
        steps.append([int(np.round(actual_step_size * i)) for i in range(num_steps)])
    return steps


@torch.no_grad()
def sliding_window_predict(net, image, patch_size, step_size, return_probabilities=False):
    """image: [C, D, H, W] (numpy or tensor, any device) with every extent >= the patch extent.  Returns the uint8
    label volume [D, H, W] on the model's device (and the mean class probabilities when asked)."""
    dev = next(net.parameters()).device
    if dev.type != "cuda":
        raise _lib.HdfError("sliding_window_predict needs the model on a GPU (there is no CPU path)")
    image = torch.as_tensor(image).float().to(dev)
    size = tuple(int(s) for s in image.shape[1:])
    patch = tuple(int(p) for p in patch_size)
    if any(s < p for s, p in zip(size, patch)):
        # the reference would feed a smaller-than-patch window to the net (trainer.py:536-547); the plan here is built
        # for one window size
        raise ValueError(f"volume {size} is smaller than the patch {patch}: pad it or pick a smaller patch")
    was_training = net.training
    net.eval()
    n_cls = net.n_cls
    psum = torch.zeros((n_cls,) + size, device=dev)
    cnt = torch.zeros(size, device=dev)
    steps = cal_steps(size, patch, step_size)
    try:
        for x in steps[0]:
            for y in steps[1]:
                for z in steps[2]:
                    data = image[None, :, x:x + patch[0], y:y + patch[1], z:z + patch[2]].contiguous()
                    logits = net(data)[0]
                    if logits.dtype == torch.bfloat16:
                        dt = _lib.BF16
                    elif logits.dtype == torch.float32:
                        dt = _lib.F32
                    else:
                        raise _lib.HdfError(f"unsupported logits dtype {logits.dtype}")
                    logits = logits.contiguous()
                    check(lib().hdf_sw_accumulate(dt, ptr(logits), n_cls, patch[0], patch[1], patch[2], ptr(psum),
                                                  ptr(cnt), size[0], size[1], size[2], x, y, z, stream_ptr()),
                          "hdf_sw_accumulate")
    finally:
        net.train(was_training)
    label = torch.empty(size, dtype=torch.uint8, device=dev)
    check(lib().hdf_sw_finalize(ptr(psum), ptr(cnt), n_cls, psum[0].numel(), ptr(label), stream_ptr()), "hdf_sw_finalize")
    if return_probabilities:
        return label, psum / cnt
    return label


def onehot_from_labels(labels, n_cls):
    """uint8 class map [N, D, H, W] (device tensor) -> fp32 one-hot [N, n_cls, D, H, W], the To_Tensor layout of
    data_utils/data_loader.py:146-151 (values >= n_cls count as background)."""
    if labels.device.type != "cuda":
        raise _lib.HdfError("onehot_from_labels needs a device tensor (there is no CPU path)")
    labels = labels.to(torch.uint8).contiguous()
    n = labels.shape[0]
    vox = labels[0].numel()
    out = torch.empty((n, n_cls) + tuple(labels.shape[1:]), dtype=torch.float32, device=labels.device)
    check(lib().hdf_onehot_from_labels(ptr(labels), ptr(out), n, n_cls, vox, stream_ptr()), "hdf_onehot_from_labels")
    return out
