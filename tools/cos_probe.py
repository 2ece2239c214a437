"""Why does ONE conv weight gradient -- block_1_2_left.conv.weight, the second conv of the encoder's top level
(reference models/HDenseFormer.py:196-199) -- sit at cosine 0.950 against the fp32 run when every other tensor of the bf16
step is >= 0.98?  (VERDICT r05 #7.)  The tool separates the kernel's arithmetic from its operands.  A g4-geometry step
(n_filters 32, 64^3, batch 2, train mode) is run with fp32 storage and with bf16 storage; for that layer it takes
     dy  = the gradient w.r.t. the conv's raw output (workspace buffer g.y_0 after the backward) and
     x   = relu(InstanceNorm(raw output of block_1_1_left)) = the conv's input (buffer y.block_1_1_left + its statistics)
from both runs and forms the weight gradient  dW[o][i][tap] = sum_v dy[v][o] x[v + tap - 1][i]  with torch in float64 from
every combination of the two runs' operands.  Prints the cosines; one JSON line at the end."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "h-denseformer_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import torch.nn.functional as F

from oracle import detgen
from oracle import hdf_oracle as orc

DEV = "cuda:0"
LAYER, PREV = "block_1_2_left", "block_1_1_left"


def cos(a, b):
    a, b = a.flatten().double(), b.flatten().double()
    return float((a @ b) / (a.norm() * b.norm() + 1e-300))


def run(cfg, batch, dtype, seed=11):
    from loss.combine_loss import CEPlusDice, DeepSuperloss
    from models.HDenseFormer import HDenseFormer
    in_ch, n_cls, nf, size, td = cfg
    net = HDenseFormer(in_ch, n_cls, nf, image_size=size, transformer_depth=td)
    sd = orc.det_model(*cfg)
    net.load_state_dict(sd)
    net = net.to(DEV)
    net.compute_dtype = dtype
    net.train()
    net.set_dropout_seed(seed)
    x = torch.from_numpy(detgen.det_input(batch, in_ch, size, tag="cos_probe")).to(DEV)
    onehot = torch.from_numpy(detgen.one_hot(detgen.det_labels(batch, n_cls, size, tag="cos_probe"), n_cls)).to(DEV)
    crit = DeepSuperloss(criterion=CEPlusDice(weight=None, ignore_index=0))
    crit(net(x), onehot).backward()
    torch.cuda.synchronize()
    rt = net._last_rt
    grads = {k: p.grad.detach().double().clone() for k, p in net.named_parameters()}
    dy = rt.read_buffer("g.y_0").double()                       # [B, C, D, H, W]
    yprev = rt.read_buffer("y." + PREV).double()
    params = dict(net.named_parameters())
    gamma, beta = params[PREV + ".norm.weight"].detach().double(), params[PREV + ".norm.bias"].detach().double()
    mu = yprev.mean(dim=(2, 3, 4), keepdim=True)
    var = yprev.var(dim=(2, 3, 4), unbiased=False, keepdim=True)
    xin = torch.relu((yprev - mu) / torch.sqrt(var + 1e-5) * gamma.view(1, -1, 1, 1, 1) + beta.view(1, -1, 1, 1, 1))
    if dtype != "fp32":   # what the kernels see: the transformed input rounded to the storage type on its way into LDS
        xin = xin.to(torch.bfloat16).double()
    # the InstanceNorm(+ReLU) backward of LAYER itself, in float64, from what the run stored: da = d(ds_0) (buffer g.ds0, the
    # gradient of relu(IN(y))), y = the conv's raw output
    da = rt.read_buffer("g.ds0").double()
    y = rt.read_buffer("y." + LAYER).double()
    g2, b2 = params[LAYER + ".norm.weight"].detach().double(), params[LAYER + ".norm.bias"].detach().double()
    mu2 = y.mean(dim=(2, 3, 4), keepdim=True)
    rstd2 = 1.0 / torch.sqrt(y.var(dim=(2, 3, 4), unbiased=False, keepdim=True) + 1e-5)
    xh = (y - mu2) * rstd2
    gg = da * ((xh * g2.view(1, -1, 1, 1, 1) + b2.view(1, -1, 1, 1, 1)) > 0)
    m1 = gg.mean(dim=(2, 3, 4), keepdim=True)
    m2 = (gg * xh).mean(dim=(2, 3, 4), keepdim=True)
    k1 = g2.view(1, -1, 1, 1, 1) * rstd2
    dy_f64 = k1 * (gg - m1 - xh * m2)
    # d(ds_0) = [gradient routed back through MaxPool3d(2)] + [the decoder's skip-path gradient].  The first part is
    # g.pool1 scattered to the arg-max voxel of every 2x2x2 window of ds_0 (this run's OWN stored ds_0 decides the voxel)
    ds0 = rt.read_buffer("ds0").double()
    dpool = rt.read_buffer("g.pool1").double()
    _, idx = F.max_pool3d(ds0.float(), 2, return_indices=True)
    pool_part = F.max_unpool3d(dpool.float(), idx, 2).double()
    extra = {"da": da, "dy_f64": dy_f64, "cancel": float(dy_f64.norm() / (k1 * gg).norm()),
             "pool_part": pool_part, "skip_part": da - pool_part, "idx": idx, "dpool": dpool,
             "mean_share": float((k1 * m1.expand_as(gg)).norm() / (k1 * gg).norm())}
    return grads, dy, xin, extra


def wgrad(dy, x):
    """dW [Cout][Cin][3][3][3] in float64 (the definition, via autograd of conv3d)"""
    for dt in (torch.float64, torch.float32):       # (float32 only if this torch build has no float64 conv on the device)
        try:
            w = torch.zeros(dy.shape[1], x.shape[1], 3, 3, 3, dtype=dt, device=x.device, requires_grad=True)
            out = F.conv3d(x.to(dt), w, padding=1)
            (g,) = torch.autograd.grad(out, w, dy.to(dt))
            return g.double()
        except RuntimeError as e:
            print("wgrad in", dt, "failed:", str(e)[:200])
    raise SystemExit("no conv3d available for the probe")


def main():
    cfg, batch = (4, 4, 32, (64, 64, 64), 8), 2
    g32, dy32, x32, e32 = run(cfg, batch, "fp32")
    g16, dy16, x16, e16 = run(cfg, batch, "bf16")
    name = LAYER + ".conv.weight"
    rec = {"layer": name, "cos_step_bf16_vs_fp32": cos(g16[name], g32[name]),
           "cos_dy": cos(dy16, dy32), "cos_x": cos(x16, x32),
           "rel_l2_dy": float((dy16 - dy32).norm() / dy32.norm()), "rel_l2_x": float((x16 - x32).norm() / x32.norm())}
    # where dy's error comes from: the stored d(activation) of the two runs, the InstanceNorm backward redone in float64
    # from each run's stored tensors, and how much of |k1 g| survives the subtraction of the two means
    rec["cos_da"] = cos(e16["da"], e32["da"])
    rec["rel_l2_da"] = float((e16["da"] - e32["da"]).norm() / e32["da"].norm())
    rec["cos_f64_in_bwd_of_bf16_tensors_vs_bf16_dy"] = cos(e16["dy_f64"], dy16)
    rec["cos_f64_in_bwd_of_bf16_tensors_vs_fp32_dy"] = cos(e16["dy_f64"], dy32)
    rec["cos_f64_in_bwd_of_fp32_tensors_vs_fp32_dy"] = cos(e32["dy_f64"], dy32)
    rec["norm_dy_over_norm_k1_g"] = e32["cancel"]
    rec["norm_mean_term_over_norm_k1_g"] = e32["mean_share"]
    # ... and which of the two parts of d(ds_0) differs: arg-max decisions of the pooling windows that changed between the
    # runs (values that are distinct in fp32 and equal, or swapped, after rounding ds_0 to bf16) move a window's whole
    # gradient to another voxel
    live = e32["dpool"].abs() > 0
    rec["pool_windows_with_another_argmax"] = float(((e16["idx"] != e32["idx"]) & live).sum() / live.sum())
    rec["cos_pool_part"] = cos(e16["pool_part"], e32["pool_part"])
    rec["cos_skip_part"] = cos(e16["skip_part"], e32["skip_part"])
    rec["cos_pooled_gradient_g_pool1"] = cos(e16["dpool"], e32["dpool"])
    rec["pool_part_energy_share"] = float(e32["pool_part"].norm() ** 2 / e32["da"].norm() ** 2)
    # the fp32 run's pooled gradient routed by the BF16 run's arg-max decisions: isolates the decisions from the values
    mixed = F.max_unpool3d(e32["dpool"].float(), e16["idx"], 2).double()
    rec["cos_fp32_pooled_gradient_routed_by_bf16_argmax"] = cos(mixed, e32["pool_part"])
    combos = {"dy32_x32": (dy32, x32), "dy16_x16": (dy16, x16), "dy16_x32": (dy16, x32), "dy32_x16": (dy32, x16)}
    dw = {k: wgrad(a, b) for k, (a, b) in combos.items()}
    rec["cos_f64_from_fp32_operands_vs_fp32_step"] = cos(dw["dy32_x32"], g32[name])     # sanity: the probe reads the right buffers
    rec["cos_f64_from_bf16_operands_vs_bf16_step"] = cos(dw["dy16_x16"], g16[name])     # the kernel's own arithmetic
    rec["cos_f64_from_bf16_operands_vs_fp32_step"] = cos(dw["dy16_x16"], g32[name])     # what the OPERANDS alone cost
    rec["cos_f64_bf16_dy_fp32_x_vs_fp32_step"] = cos(dw["dy16_x32"], g32[name])
    rec["cos_f64_fp32_dy_bf16_x_vs_fp32_step"] = cos(dw["dy32_x16"], g32[name])
    # how much of dW survives the cancellation in the voxel sum: |sum| / sum|.| per entry (a small ratio = the entry is
    # the difference of large terms, so a relative error of 2^-9 per term is a large relative error of the sum)
    absw = wgrad(dy32.abs(), x32.abs())
    rec["median_cancellation_ratio"] = float((dw["dy32_x32"].abs() / (absw + 1e-300)).median())
    other = "block_1_2_right.conv.weight"
    rec["cos_step_other_layer_" + other] = cos(g16[other], g32[other])
    for k, v in rec.items():
        print(f"{k:55s} {v}")
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
