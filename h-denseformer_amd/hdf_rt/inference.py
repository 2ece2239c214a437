"""Sliding-window inference on the device (reference: SemanticSeg.inference_slidingwindow / cal_steps,
trainer.py:488-618).  The reference runs every window through the net, takes the softmax on the GPU, then adds into two
full-size fp32 accumulators with indexed torch ops and finally argmaxes a softmax of their quotient on the GPU and
moves it to the host.  Here the per-window tail (softmax + accumulate + count) is ONE HIP kernel reading the logits
once, the vote is one kernel writing a uint8 map; the model is the MI355X-native HDenseFormer.  No CPU fallback."""
import numpy as np
import torch

from . import _lib
from ._lib import check, lib, ptr, stream_ptr

_SW_DT = {torch.float32: _lib.F32, torch.bfloat16: _lib.BF16, torch.float16: _lib.F16}


def cal_steps(image_size, patch_size, step_size):
    """Window origins along every axis (reference: SemanticSeg.cal_steps, trainer.py:595-618): an axis no longer than
    the patch gets the single origin 0; otherwise the span `size - patch` is covered by the fewest windows whose
    spacing does not exceed `step`, spread evenly (rounded), so that the last window ends at the volume's end."""
    origins = []
    for size, patch, step in zip(image_size, patch_size, step_size):
        span = size - patch
        if span <= 0:
            origins.append([0])
            continue
        n = int(np.ceil(span / step)) + 1
        origins.append([int(np.round(span / (n - 1) * k)) for k in range(n)])
    return origins


@torch.no_grad()
def sliding_window_predict(net, image, patch_size, step_size, return_probabilities=False, window_batch=4):
    """image: [C, D, H, W] (numpy or tensor, any device).  Returns the uint8 label volume [D, H, W] on the model's
    device (and the mean class probabilities when asked).

    Reference loop: trainer.py:527-584.  Windows are gathered `window_batch` at a time into ONE forward of the plan
    (the reference runs them one by one).  An axis shorter than the patch yields the reference's clipped window
    (trainer.py:529-541: `ub = x + patch if it fits else the volume's end`); the reference would hand that smaller
    tensor to the net, which HDenseFormer itself cannot take (its position embeddings fix the token grid), so the
    window is zero-padded up to the patch, run, and only the logits over the real voxels are accumulated."""
    dev = next(net.parameters()).device
    if dev.type != "cuda":
        raise _lib.HdfError("sliding_window_predict needs the model on a GPU (there is no CPU path)")
    image = torch.as_tensor(image).float().to(dev)
    if image.dim() != 4:
        raise ValueError(f"image must be [C, D, H, W], got {tuple(image.shape)}")
    size = tuple(int(s) for s in image.shape[1:])
    patch = tuple(int(p) for p in patch_size)
    if patch != tuple(net.image_size):
        raise ValueError(f"patch {patch} differs from the model's image_size {tuple(net.image_size)}")
    ext = tuple(min(s, p) for s, p in zip(size, patch))          # clipped window extent (trainer.py:529-541)
    was_training = net.training
    net.eval()
    n_cls = net.n_cls
    psum = torch.zeros((n_cls,) + size, device=dev)
    cnt = torch.zeros(size, device=dev)
    steps = cal_steps(size, patch, step_size)
    origins = [(x, y, z) for x in steps[0] for y in steps[1] for z in steps[2]]
    wb = max(1, int(window_batch))
    try:
        for i0 in range(0, len(origins), wb):
            group = origins[i0:i0 + wb]
            if ext == patch:
                data = torch.stack([image[:, x:x + patch[0], y:y + patch[1], z:z + patch[2]] for x, y, z in group])
            else:
                data = torch.zeros((len(group), image.shape[0]) + patch, device=dev)
                for k, (x, y, z) in enumerate(group):
                    data[k, :, :ext[0], :ext[1], :ext[2]] = image[:, x:x + ext[0], y:y + ext[1], z:z + ext[2]]
            logits = net(data.contiguous())[0]
            if logits.dtype not in _SW_DT:
                raise _lib.HdfError(f"unsupported logits dtype {logits.dtype}")
            if ext != patch:
                logits = logits[:, :, :ext[0], :ext[1], :ext[2]]
            logits = logits.contiguous()
            per = logits[0].numel() * logits.element_size()
            for k, (x, y, z) in enumerate(group):
                check(lib().hdf_sw_accumulate(_SW_DT[logits.dtype], ptr(logits) + k * per, n_cls, ext[0], ext[1],
                                              ext[2], ptr(psum), ptr(cnt), size[0], size[1], size[2], x, y, z,
                                              stream_ptr()), "hdf_sw_accumulate")
    finally:
        net.train(was_training)
    label = torch.empty(size, dtype=torch.uint8, device=dev)
    check(lib().hdf_sw_finalize(ptr(psum), ptr(cnt), n_cls, psum[0].numel(), ptr(label), stream_ptr()), "hdf_sw_finalize")
    if return_probabilities:
        return label, psum / cnt
    return label


def _normalize(image, mode, mean=0.0, w=1024.0):
    if not torch.is_tensor(image) or image.device.type != "cuda":
        raise _lib.HdfError("device-side normalisation needs a GPU tensor (there is no CPU path)")
    if image.dtype != torch.float32 or image.dim() < 2:
        raise ValueError("image must be a float32 tensor [C, ...]")
    img = image.contiguous()
    c, vox = img.shape[0], img[0].numel()
    ws = torch.empty(lib().hdf_normalize_workspace_bytes(c), dtype=torch.uint8, device=img.device)
    if mode == 0:
        check(lib().hdf_normalize_mr(ptr(img), c, vox, ptr(ws), stream_ptr()), "hdf_normalize_mr")
    else:
        check(lib().hdf_normalize_petct(ptr(img), c, vox, float(mean), float(w), ptr(ws), stream_ptr()),
              "hdf_normalize_petct")
    if img.data_ptr() != image.data_ptr():
        image.copy_(img)
    return image


def mr_normalize_(image):
    """MRNormalize (data_utils/data_loader.py:39-50) in place on a device sample [C, D, H, W]: the raw volume crosses
    PCIe once and is scaled where the step consumes it."""
    return _normalize(image, 0)


def pet_ct_normalize_(image, mean=0, w=1024):
    """PETandCTNormalize (data_utils/data_loader.py:53-68) in place on a device sample [2+, D, H, W]."""
    return _normalize(image, 1, mean, w)


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
