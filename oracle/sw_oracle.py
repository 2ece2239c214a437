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
