"""Golden vectors for SURVEY.md 8(f) rows 2 and 3: sliding-window inference and label staging.

TEST INFRASTRUCTURE, build container only (imports the REAL reference from /root/reference).
Usage:   python -m oracle.make_goldens_sw

  g8_sliding_window   reference `SemanticSeg.cal_steps` (trainer.py:595-618) on several shapes, and the window loop of
                      `inference_slidingwindow` (trainer.py:527-580: net(patch)[0] -> softmax -> += into the window,
                      count += 1, output = argmax(softmax(sum / count))) run with the reference model
                      HDenseFormer(2,3,16,(32,)*3,td=8) on a 2x48x40x32 volume: argmax map + strided mean probabilities
  g10_normalize       reference `MRNormalize` / `PETandCTNormalize` (data_utils/data_loader.py:39-68) on seeded volumes
                      (one all-zero channel, negative values, values beyond the CT clip window)
  g9_to_tensor        reference `To_Tensor` (data_utils/data_loader.py:126-159) one-hot of a uint8 label map that also
                      holds values >= n_cls (they fall into the background channel)
"""
import json
import os
import sys

sys.dont_write_bytecode = True

import numpy as np
import torch
import torch.nn.functional as F

from . import detgen
from .make_goldens import OUT, _import_reference, _load

CFG = (2, 3, 16, (32, 32, 32), 8)
IMAGE = (48, 40, 32)
STEP = (16, 16, 16)


def main():
    ref = _import_reference()
    trainer = ref["trainer"]
    seg = trainer.SemanticSeg.__new__(trainer.SemanticSeg)
    cases = [((32, 32, 32), (16, 16, 16), (48, 40, 32)), ((128, 128, 128), (64, 64, 64), (155, 240, 240)),
             ((96, 96, 96), (48, 48, 48), (96, 200, 97)), ((64, 64, 32), (32, 32, 16), (300, 64, 33))]
    steps_out = []
    for patch, step, size in cases:
        seg.patch_size, seg.step_size = patch, step
        steps_out.append({"patch": patch, "step": step, "size": size, "steps": seg.cal_steps(size)})

    in_ch, n_cls, nf, patch, td = CFG
    net = ref["HDenseFormer"](in_ch, n_cls, nf, image_size=patch, transformer_depth=td)
    _load(net, CFG)
    net.eval()
    image = detgen.det_input(1, in_ch, IMAGE, tag="sw")[0]                      # [C, D, H, W]
    seg.patch_size, seg.step_size = patch, STEP
    steps = seg.cal_steps(image.shape[1:])
    agg = torch.zeros((1, n_cls) + IMAGE)
    cnt = torch.zeros((1, n_cls) + IMAGE)
    with torch.no_grad():
        for x in steps[0]:
            ub_x = x + patch[0] if x + patch[0] <= IMAGE[0] else IMAGE[0]
            for y in steps[1]:
                ub_y = y + patch[1] if y + patch[1] <= IMAGE[1] else IMAGE[1]
                for z in steps[2]:
                    ub_z = z + patch[2] if z + patch[2] <= IMAGE[2] else IMAGE[2]
                    data = torch.from_numpy(image[:, x:ub_x, y:ub_y, z:ub_z][None]).float()
                    pred = net(data)[0].float()
                    pred = F.softmax(pred, dim=1)
                    pred = F.interpolate(pred, (ub_x - x, ub_y - y, ub_z - z))
                    agg[:, :, x:ub_x, y:ub_y, z:ub_z] += pred
                    cnt[:, :, x:ub_x, y:ub_y, z:ub_z] += 1
    mean = agg / cnt
    out = torch.argmax(torch.softmax(mean, dim=1), 1).numpy().squeeze().astype(np.uint8)
    np.savez_compressed(os.path.join(OUT, "g8_sliding_window.npz"), cfg=np.array([in_ch, n_cls, nf, td]),
                        patch=np.array(patch), step=np.array(STEP), image_size=np.array(IMAGE),
                        steps_json=np.array(json.dumps(steps_out)), window_steps=np.array(json.dumps(steps)),
                        argmax=out, mean_s2=mean[0, :, ::2, ::2, ::2].numpy(), count_s2=cnt[0, 0, ::2, ::2, ::2].numpy(),
                        torch_version=np.array(torch.__version__))
    print("g8_sliding_window: windows", [len(s) for s in steps], "classes", np.bincount(out.ravel(), minlength=n_cls))

    # ---- To_Tensor
    sys.path.insert(0, "/root/reference")
    from data_utils.data_loader import To_Tensor
    lab = (detgen.mix32(np.arange(12 * 10 * 8, dtype=np.uint32) + np.uint32(77)) % 6).astype(np.uint8).reshape(12, 10, 8)
    img = np.zeros((4, 12, 10, 8), dtype=np.float32)
    s = To_Tensor(num_class=4, input_channel=4)({"image": img, "label": lab})
    np.savez_compressed(os.path.join(OUT, "g9_to_tensor.npz"), label=lab, onehot=s["label"].numpy(), n_cls=np.array(4))
    print("g9_to_tensor: label values", np.unique(lab), "background voxels", int(s["label"][0].sum()))

    # ---- MRNormalize / PETandCTNormalize
    from data_utils.data_loader import MRNormalize, PETandCTNormalize
    shape = (4, 12, 20, 24)
    n = int(np.prod(shape))
    mr = ((detgen._uniform("g10.mr", n) * 900.0 + 700.0).astype(np.float32)).reshape(shape)    # mostly positive, some < 0
    mr[2] = 0.0                                                                                # all-zero channel: untouched
    mr_out = MRNormalize()({"image": mr.copy(), "label": None})["image"]
    pc = np.stack([(detgen._uniform("g10.ct", n // 4) * 1800.0).astype(np.float32),           # beyond +-1024: clipped
                   (detgen._uniform("g10.pet", n // 4) * 3.0 + 5.0).astype(np.float32)]).reshape((2,) + shape[1:])
    pc_out = PETandCTNormalize()({"image": pc.copy(), "label": None})["image"]
    pc2_out = PETandCTNormalize(mean=40, w=400)({"image": pc.copy(), "label": None})["image"]
    np.savez_compressed(os.path.join(OUT, "g10_normalize.npz"), mr_in=mr, mr_out=mr_out, petct_in=pc, petct_out=pc_out,
                        petct_m40_w400_out=pc2_out)
    print("g10_normalize: mr range", mr_out.min(), mr_out.max(), "petct ch0 range", pc_out[0].min(), pc_out[0].max(),
          "ch1 mean/std", pc_out[1].mean(), pc_out[1].std())


if __name__ == "__main__":
    main()
