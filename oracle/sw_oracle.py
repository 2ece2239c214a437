"""CPU restatement of the sliding-window inference loop and of the label staging.

TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this package.
Pinned against goldens captured from the real reference: tests/golden/g8_sliding_window.npz, g9_to_tensor.npz
(oracle/make_goldens_sw.py)."""
import numpy as np
import torch


def cal_steps(image_size, patch_size, step_size):
    """trainer.py:595-618: window origins per axis; the last window ends exactly at the volume's end."""
    steps = []
    for dim in range(len(image_size)):
        if image_size[dim] <= patch_size[dim]:
            steps.append([0])
            continue
        max_step_value = image_size[dim] - patch_size[dim]
        num_steps = int(np.ceil(max_step_value / step_size[dim])) + 1
        actual = max_step_value / (num_steps - 1)
        steps.append([int(np.round(actual * i)) for i in range(num_steps)])
    return steps


def sliding_window(forward, image, n_cls, patch_size, step_size):
    """trainer.py:527-584 with `forward(patch[1,C,pd,ph,pw]) -> logits[1,n_cls,pd,ph,pw]` in place of net(data)[0]:
    softmax per window, sum and count per voxel, label = argmax(softmax(sum / count)).  Returns (labels uint8
    [D,H,W], mean probabilities [n_cls,D,H,W])."""
    image = torch.as_tensor(image).float()
    size = tuple(image.shape[1:])
    steps = cal_steps(size, patch_size, step_size)
    agg = torch.zeros((n_cls,) + size)
    cnt = torch.zeros(size)
    for x in steps[0]:
        ux = min(x + patch_size[0], size[0])
        for y in steps[1]:
            uy = min(y + patch_size[1], size[1])
            for z in steps[2]:
                uz = min(z + patch_size[2], size[2])
                logits = forward(image[None, :, x:ux, y:uy, z:uz])
                agg[:, x:ux, y:uy, z:uz] += torch.softmax(logits.float()[0], dim=0)
                cnt[x:ux, y:uy, z:uz] += 1
    mean = agg / cnt
    lab = torch.argmax(torch.softmax(mean, dim=0), 0).numpy().astype(np.uint8)
    return lab, mean.numpy()


def to_onehot(label, n_cls):
    """data_loader.py:146-151 (To_Tensor): channel z >= 1 is (label == z); channel 0 is 'no other class'."""
    label = np.asarray(label)
    out = np.empty((n_cls,) + label.shape, dtype=np.float32)
    for z in range(1, n_cls):
        out[z] = (label == z).astype(np.float32)
    out[0] = np.amax(out[1:], axis=0) == 0
    return out


def mr_normalize(image):
    """data_loader.py:39-50 (MRNormalize): per channel divide by the channel maximum when it is non-zero, then clamp
    negatives to 0.  fp32 arithmetic like the reference's numpy."""
    image = np.array(image, dtype=np.float32, copy=True)
    for i in range(image.shape[0]):
        m = np.max(image[i])
        if m != 0:
            image[i] = image[i] / m
    image[image < 0] = 0
    return image


def pet_ct_normalize(image, mean=0, w=1024):
    """data_loader.py:53-68 (PETandCTNormalize): channel 0 clipped to mean +- w and mapped to [-1, 1]; channel 1
    z-scored with the population std + 1e-3."""
    image = np.array(image, dtype=np.float32, copy=True)
    image[0] = (np.clip(image[0], mean - w, mean + w) - mean) / w
    mu, sd = np.mean(image[1]), np.std(image[1])
    image[1] = (image[1] - mu) / (sd + 1e-3)
    return image
