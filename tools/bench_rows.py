"""Measurement of the SURVEY 8f rows on one MI355X (HIP-event medians; inputs resident on the device), each next to a
bounded host-side restatement of what the reference does for the same call (numpy / torch CPU on this box's cores).

  python tools/bench_rows.py > profiles/<round>_rows.json      (one JSON object per line)
"""
import json
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "h-denseformer_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch

from hdf_rt import inference
from hdf_rt.loss_fn import RunningDice, compute_dice
from hdf_rt.optim import FlatAdam

DEV = torch.device("cuda", 0)


def gpu_ms(fn, reps=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]


def cpu_s(fn, reps=3):
    fn()
    ts = []
    for _ in range(reps):
        t = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t)
    return sorted(ts)[len(ts) // 2]


def emit(**kw):
    print(json.dumps(kw), flush=True)


g = torch.Generator().manual_seed(7)
B, C, S = 2, 4, 128
vox = S ** 3
logits = torch.randn(B, C, S, S, S, generator=g).to(DEV).to(torch.bfloat16)
lab = torch.randint(0, C, (B, S, S, S), generator=g, dtype=torch.uint8)
onehot = torch.nn.functional.one_hot(lab.long(), C).permute(0, 4, 1, 2, 3).float().contiguous().to(DEV)
lab_d = lab.to(DEV)

# ---- 8f-1: step metrics
ms = gpu_ms(lambda: compute_dice(logits, onehot))
by = logits.numel() * 2 + onehot.numel() * 4
lg_c, oh_c = logits[:1].float().cpu(), onehot[:1].cpu()


def ref_dice():   # trainer.py:919-945 on the host, one sample: softmax, argmax, per-class overlap
    p = torch.softmax(lg_c, 1).argmax(1)
    t = oh_c.argmax(1)
    return [float(2 * ((p == c) & (t == c)).sum() / (((p == c).sum() + (t == c).sum()).clamp(min=1))) for c in range(1, C)]


emit(row="8f-1 compute_dice", config=f"{B}x{C}x{S}^3 bf16 logits + fp32 one-hot", gpu_ms=ms, algorithmic_bytes=by,
     achieved_GBps=by / ms / 1e6, hbm_peak_GBps=8000, cpu_baseline={"kind": "port", "what": "torch CPU softmax+argmax+overlap, one sample",
                                                                     "s_per_sample": cpu_s(ref_dice), "threads": torch.get_num_threads()})
rd = RunningDice(list(range(C)), ignore_label=-1)
ms = gpu_ms(lambda: rd.update_from_logits(onehot, logits))
gt_np, pr_np = lab[:1].numpy().ravel(), lg_c.argmax(1).numpy().ravel().astype(np.uint8)
emit(row="8f-1 RunningDice.update_matrix", config=f"{B}x{C}x{S}^3 (one-hot, logits) -> {C}x{C} counts", gpu_ms=ms,
     algorithmic_bytes=by, achieved_GBps=by / ms / 1e6, hbm_peak_GBps=8000,
     cpu_baseline={"kind": "port", "what": "numpy bincount confusion matrix of two label maps, one sample (metrics.py:104-133 without the D2H)",
                   "s_per_sample": cpu_s(lambda: np.bincount(gt_np.astype(np.int64) * C + pr_np, minlength=C * C)), "threads": 1})

# ---- 8f-3: staging
ms = gpu_ms(lambda: inference.onehot_from_labels(lab_d, C))
by = lab_d.numel() + lab_d.numel() * C * 4
emit(row="8f-3 onehot_from_labels", config=f"{B}x{S}^3 uint8 -> fp32 one-hot", gpu_ms=ms, algorithmic_bytes=by,
     achieved_GBps=by / ms / 1e6, hbm_peak_GBps=8000,
     cpu_baseline={"kind": "port", "what": "numpy one-hot of one sample (data_loader.py:146-151)", "threads": 1,
                   "s_per_sample": cpu_s(lambda: np.stack([(lab[0].numpy() == c) for c in range(C)]).astype(np.float32))})
vol = torch.rand(C, S, S, S, generator=g).to(DEV) * 900
ms = gpu_ms(lambda: inference.mr_normalize_(vol.clone()))
ms0 = gpu_ms(lambda: vol.clone())
by = vol.numel() * 4 * 3      # statistics pass + read + write
vol_np = vol.cpu().numpy()


def ref_mr():
    out = np.empty_like(vol_np)
    for c in range(C):
        v = vol_np[c]
        out[c] = (v - v.min()) / max(v.max() - v.min(), 1e-8)
    return out


emit(row="8f-3 mr_normalize_", config=f"{C}x{S}^3 fp32 in place (clone time {ms0:.3f} ms subtracted)", gpu_ms=ms - ms0,
     algorithmic_bytes=by, achieved_GBps=by / max(ms - ms0, 1e-6) / 1e6, hbm_peak_GBps=8000,
     cpu_baseline={"kind": "port", "what": "numpy per-channel min/max rescale of the same volume", "s_per_sample": cpu_s(ref_mr),
                   "threads": 1})

# ---- 8f-2: sliding-window inference
from models.HDenseFormer import HDenseFormer
net = HDenseFormer(4, 4, 32, image_size=(128, 128, 128), transformer_depth=24).to(DEV)
net.compute_dtype = "bf16"
net.eval()
big = torch.rand(4, 192, 192, 160, generator=g).to(DEV)
steps = inference.cal_steps(tuple(big.shape[1:]), (128, 128, 128), (64, 64, 64))
nwin = len(steps[0]) * len(steps[1]) * len(steps[2])
for wb in (1, 4):
    ms = gpu_ms(lambda: inference.sliding_window_predict(net, big, (128, 128, 128), (64, 64, 64), window_batch=wb), reps=3, warm=1)
    emit(row="8f-2 sliding_window_predict", config=f"4x192x192x160 volume, 128^3 patch, step 64: {nwin} windows, window_batch={wb}, bf16",
         gpu_ms=ms, windows_per_s=nwin / ms * 1e3, ms_per_window=ms / nwin)
xb = torch.rand(4, 4, 128, 128, 128, generator=g).to(DEV)
with torch.no_grad():
    ms = gpu_ms(lambda: net(xb), reps=5, warm=2)
emit(row="8f-2 (forward alone)", config="4 windows of 4x128^3 in one eval forward, bf16", gpu_ms=ms, ms_per_window=ms / 4)
del net, xb, big
torch.cuda.empty_cache()

# ---- 8f-4: the 2-D model (BASELINE configs[0] geometry: 4 channels, 256^2)
from loss.combine_loss import CEPlusDice, DeepSuperloss
from models.HDenseFormer_2D import HDenseFormer_2D_32
net2 = HDenseFormer_2D_32(4, 2, (256, 256), 24).to(DEV)
net2.compute_dtype = "bf16"
x2 = torch.rand(2, 4, 256, 256, generator=g).to(DEV)
net2.eval()
with torch.no_grad():
    ms = gpu_ms(lambda: net2(x2), reps=5, warm=2)
emit(row="8f-4 HDenseFormer_2D_32 eval forward", config="batch 2 of 4x256^2, n_cls 2, td 24, bf16 (native depth-1 path, round 6)",
     gpu_ms=ms, samples_per_s=2 / ms * 1e3)
net2.train()
crit = DeepSuperloss(criterion=CEPlusDice(weight=None, ignore_index=0))
opt = FlatAdam(net2, lr=1e-3, weight_decay=1e-4)
t2 = torch.nn.functional.one_hot(torch.randint(0, 2, (2, 256, 256), generator=g), 2).permute(0, 3, 1, 2).float().contiguous().to(DEV)


def step2():
    opt.zero_grad()
    loss = crit(net2(x2), t2)
    loss.backward()
    opt.step()


ms = gpu_ms(step2, reps=5, warm=2)
emit(row="8f-4 HDenseFormer_2D_32 train step", config="batch 2 of 4x256^2, fwd + DeepSuper CE+Dice + bwd + Adam, bf16", gpu_ms=ms,
     samples_per_s=2 / ms * 1e3)

# ---- 8f-4 at the reference's own 2-D workload: PI-CAI 384^2, batch 24 (config.py:69-77), the model of test.py:4-13
del net2, opt
torch.cuda.empty_cache()
try:
    net3 = HDenseFormer_2D_32(2, 2, (384, 384), 16).to(DEV)
    net3.compute_dtype = "bf16"
    x3 = torch.rand(24, 2, 384, 384, generator=g).to(DEV)
    net3.eval()
    with torch.no_grad():
        ms = gpu_ms(lambda: net3(x3), reps=3, warm=1)
    emit(row="8f-4 HDenseFormer_2D_32 eval forward, PI-CAI shape", config="batch 24 of 2x384^2, n_cls 2, td 16, bf16 (native depth-1 path, round 6)",
         gpu_ms=ms, samples_per_s=24 / ms * 1e3)
    net3.train()
    opt3 = FlatAdam(net3, lr=1e-3, weight_decay=1e-4)
    t3 = torch.nn.functional.one_hot(torch.randint(0, 2, (24, 384, 384), generator=g), 2).permute(0, 3, 1, 2).float().contiguous().to(DEV)

    def step3():
        opt3.zero_grad()
        loss = crit(net3(x3), t3)
        loss.backward()
        opt3.step()

    ms = gpu_ms(step3, reps=3, warm=1)
    emit(row="8f-4 HDenseFormer_2D_32 train step, PI-CAI shape", config="batch 24 of 2x384^2, fwd + DeepSuper CE+Dice + bwd + Adam, bf16",
         gpu_ms=ms, samples_per_s=24 / ms * 1e3)
except Exception as exc:  # report, do not hide: the row then says why it is missing
    emit(row="8f-4 PI-CAI shape", error=repr(exc)[:300])
