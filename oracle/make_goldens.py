"""Generate tests/golden/*.npz by running the REAL reference (imported from /root/reference).

TEST INFRASTRUCTURE.  Runs ONLY in the build container (the reference does not travel to the GPU
box); its outputs -- inputs' recipe + expected outputs, pure data -- are committed under
tests/golden/.  Usage:   python -m oracle.make_goldens [--full]

What is captured:
  g1_tiny_eval    HDenseFormer(2,3,16,(32,)*3,td=8)  B=2, eval: 4 logits, intermediates' samples, loss, grads
  g1_tiny_train   same model, train mode with the hash dropout masks (seed 1234): logits, loss, grads
  g2_odd_eval     HDenseFormer(2,2,16,(48,)*3,td=4)  B=1 (3^3 = 27 tokens, odd grid): logits + loss
  g3_loss         DeepSuperloss(CEPlusDice) values + dL/dlogits on random 4-scale logits (C=3,4; absent class)
  g3w_loss_weighted  the class-weighted / ignore_index=None forms of CEPlusDice, DiceLoss, CrossentropyLoss (--only g3w)
  g5_full_eval    HDenseFormer_32(4,4,(128,)*3,td=24) B=1 eval: strided logits, stats, Dice, loss  (--full)
  g6_2d           HDenseFormer_2D_32(4,2,(256,256),24) single forward: shapes + strided logits  (config #1)
  g6_2d_train     HDenseFormer_2D(2,3,16,(64,96),td=8) B=2 train step: logits, loss, gradients, Adam-updated samples
  g7_metric       trainer.compute_dice / metrics.RunningDice on seeded cases incl. absent class
Weights/inputs are the closed-form generators of oracle/detgen.py, so nothing but (config, seed) is
needed to regenerate the inputs on the GPU box.
"""
import argparse
import os
import sys
from unittest.mock import MagicMock

sys.dont_write_bytecode = True          # the reference mount must stay untouched
REF = "/root/reference"

import numpy as np
import torch

from . import detgen
from . import hdf_oracle as orc

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _import_reference():
    if REF not in sys.path:
        sys.path.insert(0, REF)
    for n in ["tensorboardX", "torchvision", "torchvision.transforms", "setproctitle", "h5py", "skimage",
              "skimage.transform", "skimage.util", "skimage.exposure", "transforms3d", "transforms3d.euler",
              "transforms3d.affines", "SimpleITK", "cv2"]:
        sys.modules.setdefault(n, MagicMock())
    from models.HDenseFormer import HDenseFormer, HDenseFormer_32
    from models.HDenseFormer_2D import HDenseFormer_2D, HDenseFormer_2D_32
    from loss.combine_loss import CEPlusDice, DeepSuperloss
    import trainer
    import metrics
    return dict(HDenseFormer=HDenseFormer, HDenseFormer_32=HDenseFormer_32, HDenseFormer_2D_32=HDenseFormer_2D_32,
                HDenseFormer_2D=HDenseFormer_2D,
                CEPlusDice=CEPlusDice, DeepSuperloss=DeepSuperloss, trainer=trainer, metrics=metrics)


class HashDropout:
    """Replaces torch.nn.functional.dropout while the reference runs in train mode: masks come from
    detgen.dropout_keep with site ids assigned by CALL ORDER, which in the reference forward is
    (HDenseFormer.py:230 list comprehension over modalities) emb, then per block, per layer:
    to_out, ff#1 (2 sites), ff#2 (2 sites); then the block's out_layer (2 sites)."""

    def __init__(self, seed, n_blocks):
        self.seed = seed
        order = []
        for b in range(n_blocks):
            for l in range(orc.LAYERS):
                order += [(b, l, k) for k in (detgen.KIND_ATTN_OUT, detgen.KIND_FF1_A, detgen.KIND_FF1_B,
                                              detgen.KIND_FF2_A, detgen.KIND_FF2_B)]
            order += [(b, detgen.LAYER_OUT, detgen.KIND_OUT_A), (b, detgen.LAYER_OUT, detgen.KIND_OUT_B)]
        self.per_branch = [None] + order
        self.calls = 0

    def __call__(self, inp, p=0.5, training=True, inplace=False):
        if not training:
            return inp
        m, j = divmod(self.calls, len(self.per_branch))
        self.calls += 1
        site = detgen.site_emb(m) if j == 0 else detgen.site_id(m, *self.per_branch[j])
        keep = detgen.dropout_keep(self.seed, site, inp.numel(), p)
        return inp * (torch.from_numpy(keep.reshape(tuple(inp.shape))).to(inp.dtype) / (1.0 - p))


def _load(net, cfg):
    sd = orc.det_model(*cfg)
    ref_sd = net.state_dict()
    assert list(ref_sd.keys()) == list(sd.keys()), "oracle state_dict order/keys differ from the reference"
    for k in sd:
        assert tuple(ref_sd[k].shape) == tuple(sd[k].shape), k
    net.load_state_dict(sd)
    return sd


def _stats(t):
    t = t.detach().double()
    return np.array([t.mean().item(), t.std().item(), t.abs().max().item(), t.abs().mean().item()])


def _sample(t, step):
    sl = (slice(None), slice(None)) + (slice(None, None, step),) * (t.dim() - 2)
    return t.detach()[sl].contiguous().numpy()


def _hook_intermediates(net):
    """Capture the 12 named intermediates of HDenseFormer.forward via forward hooks on the modules
    that produce them (inputs of heads / pools give the post-add tensors)."""
    got = {}
    hooks = []

    def out_hook(name):
        return lambda mod, inp, out: got.__setitem__(name, out.detach())

    def in_hook(name):
        return lambda mod, inp: got.__setitem__(name, inp[0].detach())
    hooks.append(net.deep_conv.register_forward_pre_hook(in_hook("attnall")))
    hooks.append(net.deep_conv.register_forward_hook(out_hook("attnout")))
    hooks.append(net.up1.register_forward_hook(out_hook("at1")))
    hooks.append(net.up2.register_forward_hook(out_hook("at2")))
    hooks.append(net.up3.register_forward_hook(out_hook("at3")))
    hooks.append(net.pool_1.register_forward_pre_hook(in_hook("ds0")))
    hooks.append(net.pool_2.register_forward_pre_hook(in_hook("ds1")))
    hooks.append(net.pool_3.register_forward_pre_hook(in_hook("ds2")))
    hooks.append(net.conv1x1_d3.register_forward_pre_hook(in_hook("bottleneck")))
    hooks.append(net.conv1x1_d2.register_forward_pre_hook(in_hook("dec3")))
    hooks.append(net.conv1x1_d1.register_forward_pre_hook(in_hook("dec2")))
    hooks.append(net.conv1x1.register_forward_pre_hook(in_hook("dec1")))
    return got, hooks


def _grad_summary(net, n_samples=8):
    names, norms, samples = [], [], []
    for k, p in net.named_parameters():
        g = p.grad.detach().flatten()
        names.append(k)
        norms.append(g.double().norm().item())
        idx = np.linspace(0, g.numel() - 1, n_samples).astype(np.int64)
        samples.append(g[idx].numpy())
    return np.array(names), np.array(norms), np.stack(samples)


def _param_samples(net, n_samples):
    out = []
    for k, p in net.named_parameters():
        v = p.detach().flatten()
        out.append(v[np.linspace(0, v.numel() - 1, n_samples).astype(np.int64)].numpy().copy())
    return np.stack(out)


def _det_batch(name, cfg, batch):
    """closed-form input / label map for a 3-D or 2-D config (2-D: depth-1 volumes squeezed)"""
    in_ch, n_cls, nf, size, td = cfg
    vol = size if len(size) == 3 else (1,) + tuple(size)
    x = detgen.det_input(batch, in_ch, vol, tag=name)
    lab = detgen.det_labels(batch, n_cls, vol, tag=name)
    if len(size) == 2:
        x, lab = x[:, :, 0], lab[:, 0]
    return x, lab


def golden_model(ref, name, cfg, batch, train_seed=None, sample_step=1, inter_step=4, with_grads=True, full_grads=(),
                 n_samples=8, adam_step=False):
    """adam_step: also run ONE optimizer step built by the reference's own SemanticSeg._get_optimizer
    (trainer.py:793-840; Adam, lr 1e-3, weight_decay 1e-4, its two parameter groups) and record the
    updated parameters at the sampled positions."""
    in_ch, n_cls, nf, size, td = cfg
    is2d = len(size) == 2
    cls = ref["HDenseFormer_2D"] if is2d else ref["HDenseFormer"]
    net = cls(in_ch, n_cls, nf, image_size=size, transformer_depth=td)
    _load(net, cfg)
    crit = ref["DeepSuperloss"](criterion=ref["CEPlusDice"](weight=None, ignore_index=0))
    xn, lab = _det_batch(name, cfg, batch)
    x = torch.from_numpy(xn)
    onehot = torch.from_numpy(detgen.one_hot(lab[:, None], n_cls)[:, :, 0] if is2d else detgen.one_hot(lab, n_cls))
    got, hooks = _hook_intermediates(net)
    import torch.nn.functional as F
    orig = F.dropout
    if train_seed is None:
        net.eval()
    else:
        net.train()
        F.dropout = HashDropout(train_seed, td // 4)
    try:
        outs = net(x)
        loss = crit(outs, onehot)
        if with_grads:
            loss.backward()
    finally:
        F.dropout = orig
        for h in hooks:
            h.remove()
    rec = dict(cfg=np.array([in_ch, n_cls, nf, td] + list(size)), batch=batch,
               train_seed=-1 if train_seed is None else train_seed, loss=loss.item(),
               sample_step=sample_step, inter_step=inter_step, torch_version=torch.__version__)
    for i, o in enumerate(outs):
        rec[f"out{i}"] = _sample(o, max(1, sample_step >> i))
        rec[f"out{i}_stats"] = _stats(o)
    for k, v in got.items():
        rec["inter_" + k] = _sample(v, inter_step if v.shape[-1] > 8 else 1)
        rec["inter_" + k + "_stats"] = _stats(v)
    tr = ref["trainer"]
    rec["dice_rounded"] = float(tr.compute_dice(outs[0].detach(), onehot))
    rec["dice_unrounded"] = orc.compute_dice(outs[0].detach(), onehot, rounded=False)
    if with_grads:
        names, norms, samples = _grad_summary(net, n_samples)
        rec["grad_names"], rec["grad_norms"], rec["grad_samples"] = names, norms, samples
        for k in full_grads:
            rec["gradfull_" + k] = dict(net.named_parameters())[k].grad.detach().numpy()
        if adam_step:
            from types import SimpleNamespace
            rec["param_samples_before"] = _param_samples(net, n_samples)
            opt = tr.SemanticSeg._get_optimizer(SimpleNamespace(momentum=0.99, weight_decay=1e-4), "Adam", net, 1e-3)
            opt.step()
            rec["param_samples_after"] = _param_samples(net, n_samples)
            rec["adam_lr"], rec["adam_wd"] = 1e-3, 1e-4
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)
    print(f"{name}: loss={loss.item():.6f} dice={rec['dice_rounded']:.4f}")


def golden_loss(ref):
    crit = ref["DeepSuperloss"](criterion=ref["CEPlusDice"](weight=None, ignore_index=0))
    rec = {}
    for tag, (c, s, absent) in dict(c3=(3, 16, False), c4=(4, 16, False), c4_absent=(4, 16, True)).items():
        g = torch.Generator().manual_seed(7)
        outs = [(torch.randn(2, c, s >> i, s >> i, s >> i, generator=g) * 2.0).requires_grad_(True) for i in range(4)]
        lab = torch.randint(0, c - 1 if absent else c, (2, s, s, s), generator=g)
        onehot = torch.nn.functional.one_hot(lab, c).permute(0, 4, 1, 2, 3).float()
        loss = crit(outs, onehot)
        loss.backward()
        rec[tag + "_loss"] = loss.item()
        rec[tag + "_onehot"] = onehot.numpy().astype(np.uint8)
        for i, o in enumerate(outs):
            rec[f"{tag}_logits{i}"] = o.detach().numpy()
            rec[f"{tag}_grad{i}"] = o.grad.numpy()
        print("g3", tag, loss.item())
    np.savez_compressed(os.path.join(OUT, "g3_loss.npz"), **rec)


def golden_loss_weighted(ref):
    """The class-weighted / ignore_index=None forms trainer.py:743-771 can build, from the reference's own classes."""
    from loss.cross_entropy import CrossentropyLoss
    from loss.dice_loss import DiceLoss
    cw = [0.2, 1.0, 2.5, 0.6]
    wt = torch.tensor(cw)
    rec = dict(class_weight=np.array(cw, dtype=np.float32))
    g = torch.Generator().manual_seed(17)
    c, s = 4, 16
    lab = torch.randint(0, c, (2, s, s, s), generator=g)
    onehot = torch.nn.functional.one_hot(lab, c).permute(0, 4, 1, 2, 3).float()
    rec["onehot"] = onehot.numpy().astype(np.uint8)
    base = [(torch.randn(2, c, s >> i, s >> i, s >> i, generator=g) * 2.0) for i in range(4)]
    for i, o in enumerate(base):
        rec[f"logits{i}"] = o.numpy()
    cases = dict(
        deep_w=(ref["DeepSuperloss"](criterion=ref["CEPlusDice"](weight=wt, ignore_index=0)), 4),
        cepd_w=(ref["CEPlusDice"](weight=wt, ignore_index=0), 1),
        dice_w=(DiceLoss(weight=wt, ignore_index=0, p=1), 1),            # trainer.py:761
        dice_all=(DiceLoss(weight=None, ignore_index=None), 1),
        dice_w_all=(DiceLoss(weight=wt, ignore_index=None), 1),
        ce_w=(CrossentropyLoss(weight=wt), 1))
    for tag, (crit, n) in cases.items():
        outs = [o.clone().requires_grad_(True) for o in base[:n]]
        loss = crit(outs, onehot) if n > 1 else crit(outs[0], onehot)
        loss.backward()
        rec[tag + "_loss"] = loss.item()
        for i, o in enumerate(outs):
            rec[f"{tag}_grad{i}"] = o.grad.numpy()
        print("g3w", tag, loss.item())
    np.savez_compressed(os.path.join(OUT, "g3w_loss_weighted.npz"), **rec)


def golden_metric(ref):
    tr, me = ref["trainer"], ref["metrics"]
    rec = {}
    for tag, (c, absent) in dict(c4=(4, False), c4_absent=(4, True), c3=(3, False)).items():
        g = torch.Generator().manual_seed(11)
        logits = torch.randn(2, c, 8, 8, 8, generator=g)
        lab = torch.randint(0, c - 1 if absent else c, (2, 8, 8, 8), generator=g)
        if absent:
            logits[:, c - 1] = -50.0                        # class c-1 absent from prediction too
        onehot = torch.nn.functional.one_hot(lab, c).permute(0, 4, 1, 2, 3).float()
        rec[tag + "_logits"], rec[tag + "_onehot"] = logits.numpy(), onehot.numpy().astype(np.uint8)
        rec[tag + "_dice"] = float(tr.compute_dice(logits, onehot))
        rd = me.RunningDice(labels=range(c), ignore_label=-1)
        rd.update_matrix(lab.numpy(), logits.argmax(1).numpy())
        mean, per = rd.compute_dice()
        rec[tag + "_run_dice"], rec[tag + "_run_list"] = float(mean), np.array(per, dtype=np.float64)
        print("g7", tag, rec[tag + "_dice"], mean)
    np.savez_compressed(os.path.join(OUT, "g7_metric.npz"), **rec)


def golden_2d(ref):
    cfg = (4, 2, 32, (256, 256), 24)
    net = ref["HDenseFormer_2D_32"](4, 2, (256, 256), 24)
    _load(net, cfg)
    net.eval()
    x = torch.from_numpy(detgen.det_input(1, 4, (1, 256, 256), tag="g6")[:, :, 0])
    with torch.no_grad():
        outs = net(x)
    rec = dict(cfg=np.array([4, 2, 32, 24, 256, 256]), torch_version=torch.__version__)
    for i, o in enumerate(outs):
        rec[f"shape{i}"] = np.array(o.shape)
        rec[f"out{i}"] = _sample(o, 4 if i == 0 else 1)
        rec[f"out{i}_stats"] = _stats(o)
    np.savez_compressed(os.path.join(OUT, "g6_2d.npz"), **rec)
    print("g6 shapes", [tuple(o.shape) for o in outs])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true", help="also the 4x128^3 nf32 td24 fixture (needs ~9 GB, ~1 min)")
    ap.add_argument("--only", default="", help="comma list of g1,g2,g3,g3w,g4,g5,g5t,g6,g6t,g7")
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    ref = _import_reference()
    todo = a.only.split(",") if a.only else ["g1", "g2", "g3", "g6", "g7"] + (["g5"] if a.full else [])
    if "g1" in todo:
        cfg = (2, 3, 16, (32, 32, 32), 8)
        fg = ("block_1_2_left.conv.weight", "upconv_2.weight", "conv1x1_d1.weight", "up2.double_conv.0.weight",
              "attns.1.blocks.1.0.layers.2.1.fn.to_qkv.weight", "attns.0.position_embeddings",
              "block_3_1_right.norm.weight", "attns.0.blocks.0.0.layers.0.0.weight")
        golden_model(ref, "g1_tiny_eval", cfg, 2, None, full_grads=fg)
        golden_model(ref, "g1_tiny_train", cfg, 2, 1234, full_grads=fg)
    if "g2" in todo:
        golden_model(ref, "g2_odd_eval", (2, 2, 16, (48, 48, 48), 4), 1, None, sample_step=2, inter_step=4)
    if "g3" in todo:
        golden_loss(ref)
    if "g3w" in todo:
        golden_loss_weighted(ref)
    if "g6" in todo:
        golden_2d(ref)
    if "g7" in todo:
        golden_metric(ref)
    if "g4" in todo:
        # the ws2 / wgrad2 / split-dgrad plan paths need >= 48^3 and n_filters 32: train mode, B=2, dropout on
        golden_model(ref, "g4_mid_train", (4, 4, 32, (64, 64, 64), 8), 2, 4321, sample_step=4, inter_step=8,
                     n_samples=16, adam_step=True,
                     full_grads=("conv1x1.weight", "upconv_1.bias", "upconv_2.bias", "upconv_3.bias",
                                 "block_1_1_right.norm.weight", "attns.2.blocks.1.0.layers.3.1.fn.to_qkv.weight"))
    if "g6t" in todo:
        # the 2-D model's train step (hash dropout on), gradients + one Adam step of the reference optimizer
        golden_model(ref, "g6_2d_train", (2, 3, 16, (64, 96), 8), 2, 606, sample_step=2, inter_step=4, n_samples=16,
                     adam_step=True, full_grads=("conv1x1.weight", "upconv_1.weight", "upconv_2.bias",
                                                 "block_2_1_left.conv.weight", "attns.1.patch_embeddings.weight",
                                                 "deep_conv.double_conv.0.weight"))
    if "g5t" in todo:
        # the exact computation bench.py times (BASELINE configs[1]): 4x128^3, nf32, td24, B=2, train mode
        golden_model(ref, "g5_full_train", (4, 4, 32, (128, 128, 128), 24), 2, 2024, sample_step=8, inter_step=16,
                     n_samples=16, adam_step=True,
                     full_grads=("conv1x1.weight", "upconv_1.bias", "upconv_2.bias", "upconv_3.bias",
                                 "block_1_1_right.norm.weight", "attns.3.blocks.5.0.layers.3.1.fn.to_qkv.weight"))
    if "g5" in todo:
        golden_model(ref, "g5_full_eval", (4, 4, 32, (128, 128, 128), 24), 1, None, sample_step=8, inter_step=8)


if __name__ == "__main__":
    main()
