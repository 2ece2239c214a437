"""CPU oracle: a functional restatement of the H-DenseFormer 3D training hot path.

TEST INFRASTRUCTURE ONLY -- never imported by the product package (h-denseformer_amd/).  Allowed
importers: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.

What it restates (file:line into /root/reference):
  * HDenseFormer.forward                       models/HDenseFormer.py:229-255
  * Dense_TransformerBlock.forward             models/HDenseFormer.py:132-145
  * DensePreConv_AttentionBlock.forward        models/HDenseFormer.py:91-101  (incl. the 2nd ff call :98)
  * Dense_Attention.forward                    models/HDenseFormer.py:64-75
  * DenseForward / PreNorm                     models/HDenseFormer.py:11-17,33-44
  * BasicConv3d / UpConv                       models/HDenseFormer.py:148-175
  * DeepSuperloss / CEPlusDice                 loss/combine_loss.py:8-35,68-79
  * DiceLoss / BinaryDiceLoss                  loss/dice_loss.py:5-87
  * CrossentropyLoss                           loss/cross_entropy.py:8-22
  * compute_dice / binary_dice (metric)        trainer.py:891-945
  * optimizer param grouping                   trainer.py:793-840

The arithmetic primitives of the reference live in third-party PyTorch (unpinned in the
reference's requirements.txt); the oracle uses the same torch CPU primitives
(torch.nn.functional.*) of the torch build installed in this image (2.10.0+rocm7.0, CPU kernels).
Parity pinning: tests/golden/*.npz are outputs of the REAL reference imported in the build
container (script: oracle/make_goldens.py) and tests/test_oracle_vs_golden.py checks this file
against them.  The reference itself ships no golden vectors or tests.

The model is expressed as pure functions over a state_dict-like mapping {key: tensor} with the
reference's parameter names, so no nn.Module tree is needed.  Dropout uses the counter-hash masks
of oracle/detgen.py (the HIP kernels use the same recipe) so that train-mode runs are comparable.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import detgen

HEADS = 8            # Dense_Attention heads (HDenseFormer.py:79)
GROWTH = 32          # growth_rate (HDenseFormer.py:79)
LAYERS = 4           # dense layers per block (HDenseFormer.py:79 depth=4)
PATCH = 16           # patch size (HDenseFormer.py:186)
DROP_P = 0.5         # HDenseFormer.py:79,105


class Dropper:
    """Dropout with hash masks.  seed=None -> identity (eval mode)."""

    def __init__(self, seed=None, p=DROP_P):
        self.seed, self.p = seed, p

    def __call__(self, x, site):
        if self.seed is None or self.p == 0.0:
            return x
        keep = detgen.dropout_keep(self.seed, site, x.numel(), self.p)
        m = torch.from_numpy(keep.reshape(tuple(x.shape))).to(x.dtype) / (1.0 - self.p)
        return x * m


# --------------------------------------------------------------------------- transformer branch
def _layer_norm(x, sd, p):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def _dense_forward(x, sd, p, drop, site_a, site_b):
    """Linear -> exact GELU -> dropout -> Linear -> dropout (HDenseFormer.py:36-41)."""
    h = F.gelu(F.linear(x, sd[p + ".net.0.weight"], sd[p + ".net.0.bias"]))
    h = drop(h, site_a)
    h = F.linear(h, sd[p + ".net.3.weight"], sd[p + ".net.3.bias"])
    return drop(h, site_b)


def _attention(x, sd, p, drop, site):
    """x: [B,N,32] already layer-normed.  q,k,v chunks of a bias-free 32->96 projection, 8 heads of
    width 4, scale 4**-0.5, softmax over keys, merge heads '(h d)', out projection + dropout."""
    b, n, c = x.shape
    dh = c // HEADS
    qkv = F.linear(x, sd[p + ".to_qkv.weight"])
    q, k, v = [t.reshape(b, n, HEADS, dh).permute(0, 2, 1, 3) for t in qkv.split(c, dim=-1)]
    att = torch.softmax(torch.matmul(q, k.transpose(-1, -2)) * (dh ** -0.5), dim=-1)
    o = torch.matmul(att, v).permute(0, 2, 1, 3).reshape(b, n, c)
    o = F.linear(o, sd[p + ".to_out.0.weight"], sd[p + ".to_out.0.bias"])
    return drop(o, site)


def _dense_block(x, sd, p, drop, m, bidx):
    feats = [x]
    for l in range(LAYERS):
        lp = f"{p}.layers.{l}"
        h = F.linear(torch.cat(feats, 2), sd[lp + ".0.weight"], sd[lp + ".0.bias"])
        h = _attention(_layer_norm(h, sd, lp + ".1.norm"), sd, lp + ".1.fn", drop,
                       detgen.site_id(m, bidx, l, detgen.KIND_ATTN_OUT)) + h
        h = _dense_forward(_layer_norm(h, sd, lp + ".2.norm"), sd, lp + ".2.fn", drop,
                           detgen.site_id(m, bidx, l, detgen.KIND_FF1_A),
                           detgen.site_id(m, bidx, l, detgen.KIND_FF1_B)) + h
        # the appended feature is ff evaluated AGAIN on the post-residual value (HDenseFormer.py:98)
        feats.append(_dense_forward(_layer_norm(h, sd, lp + ".2.norm"), sd, lp + ".2.fn", drop,
                                    detgen.site_id(m, bidx, l, detgen.KIND_FF2_A),
                                    detgen.site_id(m, bidx, l, detgen.KIND_FF2_B)))
    return _dense_forward(torch.cat(feats, 2), sd, p + ".out_layer", drop,
                          detgen.site_id(m, bidx, detgen.LAYER_OUT, detgen.KIND_OUT_A),
                          detgen.site_id(m, bidx, detgen.LAYER_OUT, detgen.KIND_OUT_B))


def _nd(x):
    return x.dim() - 2


def _conv(x, w, b=None, **kw):
    return (F.conv3d if _nd(x) == 3 else F.conv2d)(x, w, b, **kw)


def transformer_branch(vol, sd, m, n_blocks, drop):
    """vol: [B,1,D,H,W] one modality -> [B, 4nf, D/16, H/16, W/16]  (2D: [B,1,H,W], the
    HDenseFormer_2D variant, models/HDenseFormer_2D.py -- same graph with 2D primitives)."""
    p = f"attns.{m}"
    t = _conv(vol, sd[p + ".patch_embeddings.weight"], sd[p + ".patch_embeddings.bias"], stride=PATCH)
    b, c = t.shape[:2]
    grid = tuple(t.shape[2:])
    tok = t.flatten(2).transpose(1, 2)                       # token index = (d*gh + h)*gw + w
    tok = drop(tok + sd[p + ".position_embeddings"], detgen.site_emb(m))
    for bidx in range(n_blocks):
        tok = _dense_block(tok, sd, f"{p}.blocks.{bidx}.0", drop, m, bidx)
    # 'b (d h w) c -> b c d h w'; the trailing nearest interpolate to the same size is the identity
    return tok.transpose(1, 2).reshape(b, c, *grid)


# --------------------------------------------------------------------------------- U-Net pieces
def _basic(x, sd, p):
    """conv3 (no bias) -> InstanceNorm(affine, biased var, eps 1e-5) -> ReLU."""
    y = _conv(x, sd[p + ".conv.weight"], None, padding=1)
    return F.relu(F.instance_norm(y, weight=sd[p + ".norm.weight"], bias=sd[p + ".norm.bias"], eps=1e-5))


def _upconv(x, sd, p):
    """conv3 (+bias) -> InstanceNorm(no affine) -> ReLU -> trilinear x2 (align_corners=False)."""
    y = _conv(x, sd[p + ".double_conv.0.weight"], sd[p + ".double_conv.0.bias"], padding=1)
    y = F.relu(F.instance_norm(y, eps=1e-5))
    return F.interpolate(y, scale_factor=2, mode="trilinear" if _nd(y) == 3 else "bilinear", align_corners=False)


def _up_t(x, sd, p):
    f = F.conv_transpose3d if _nd(x) == 3 else F.conv_transpose2d
    return f(x, sd[p + ".weight"], sd[p + ".bias"], stride=2, padding=1, output_padding=1)


def _head(x, sd, p):
    return _conv(x, sd[p + ".weight"], sd[p + ".bias"])


def _pool(x):
    return F.max_pool3d(x, 2) if _nd(x) == 3 else F.max_pool2d(x, 2)


def n_blocks_of(sd, m=0):
    n = 0
    while f"attns.{m}.blocks.{n}.0.out_layer.net.0.weight" in sd:
        n += 1
    return n


def forward(x, sd, drop_seed=None, want_intermediates=False):
    """x: [B,Cin,D,H,W] fp32.  sd: mapping with the reference's state_dict keys.
    Returns [full, 1/2, 1/4, 1/8] logits (and a dict of named intermediates if asked)."""
    drop = Dropper(drop_seed)
    cin = x.shape[1]
    nb = n_blocks_of(sd)
    attnall = torch.cat([transformer_branch(x[:, m:m + 1], sd, m, nb, drop) for m in range(cin)], 1)
    attnout = _upconv(attnall, sd, "deep_conv")
    at1 = _upconv(attnout, sd, "up1")
    at2 = _upconv(at1, sd, "up2")
    at3 = _upconv(at2, sd, "up3")

    ds0 = _basic(_basic(x, sd, "block_1_1_left"), sd, "block_1_2_left") + at3
    ds1 = _basic(_basic(_pool(ds0), sd, "block_2_1_left"), sd, "block_2_2_left") + at2
    ds2 = _basic(_basic(_pool(ds1), sd, "block_3_1_left"), sd, "block_3_2_left") + at1
    bott = _basic(_basic(_pool(ds2), sd, "block_4_1_left"), sd, "block_4_2_left") + attnout

    out3 = _head(bott, sd, "conv1x1_d3")
    cat3 = torch.cat([_up_t(bott, sd, "upconv_3"), ds2], 1)
    dec3 = _basic(_basic(cat3, sd, "block_3_1_right"), sd, "block_3_2_right")
    out2 = _head(dec3, sd, "conv1x1_d2")
    cat2 = torch.cat([_up_t(dec3, sd, "upconv_2"), ds1], 1)
    dec2 = _basic(_basic(cat2, sd, "block_2_1_right"), sd, "block_2_2_right")
    out1 = _head(dec2, sd, "conv1x1_d1")
    cat1 = torch.cat([_up_t(dec2, sd, "upconv_1"), ds0], 1)
    dec1 = _basic(_basic(cat1, sd, "block_1_1_right"), sd, "block_1_2_right")
    out0 = _head(dec1, sd, "conv1x1")
    outs = [out0, out1, out2, out3]
    if want_intermediates:
        inter = dict(attnall=attnall, attnout=attnout, at1=at1, at2=at2, at3=at3, ds0=ds0, ds1=ds1,
                     ds2=ds2, bottleneck=bott, dec3=dec3, dec2=dec2, dec1=dec1, cat3=cat3, cat2=cat2, cat1=cat1)
        return outs, inter
    return outs


# ---------------------------------------------------------------------------------------- loss
def dice_term(logits, onehot, weight=None, ignore_index=0, smooth=1e-5):
    """DiceLoss (dice_loss.py:53-87): soft Dice (p=1) per class other than ignore_index, per sample then batch mean,
    times the class weight, summed and divided by C-1 (by C when ignore_index is None)."""
    c = logits.shape[1]
    prob = torch.softmax(logits, dim=1)
    dice = 0.0
    for k in range(c):
        if k == ignore_index:
            continue
        inter = (prob[:, k] * onehot[:, k]).flatten(1).sum(1)
        union = (prob[:, k] + onehot[:, k]).flatten(1).sum(1)
        term = (1.0 - (2.0 * inter + smooth) / (union + smooth)).mean()
        dice = dice + (term if weight is None else term * weight[k])
    return dice / (c - 1 if ignore_index is not None else c)


def ce_term(logits, onehot, weight=None):
    """CrossentropyLoss (cross_entropy.py:8-22): [weighted] mean CE against argmax(onehot)."""
    return F.cross_entropy(logits, onehot.argmax(1), weight=weight)


def ce_plus_dice(logits, onehot, smooth=1e-5, weight=None, ignore_index=0):
    """CEPlusDice(weight, ignore_index) (combine_loss.py:8-35): the two terms above, the class weights in both."""
    return ce_term(logits, onehot, weight) + dice_term(logits, onehot, weight, ignore_index, smooth)


def deep_super_loss(outs, onehot, weight=None, ignore_index=0):
    """sum_i 2^-i * CEPlusDice(out_i, nearest-downsampled one-hot) (combine_loss.py:72-79);
    nearest interpolation onto a 2^i-times smaller grid is the stride-2^i subsample."""
    total = 0.0
    for i, o in enumerate(outs):
        s = onehot.shape[2] // o.shape[2]
        sub = onehot[(slice(None), slice(None)) + (slice(None, None, s),) * (onehot.dim() - 2)]
        total = total + ce_plus_dice(o, sub, weight=weight, ignore_index=ignore_index) * (1.0 / (2 ** i))
    return total


# -------------------------------------------------------------------------------------- metric
def compute_dice(logits, onehot, ignore_index=0, rounded=True):
    """Hard-argmax Dice averaged over classes 1.. (trainer.py:919-945).  A class absent from both
    prediction and target keeps the value 1.0; each class value is rounded to 4 dp like the
    reference unless rounded=False."""
    pred = logits.argmax(1)
    tgt = onehot.argmax(1)
    c = onehot.shape[1]
    vals = np.ones(c, dtype=np.float32)
    for k in range(c):
        if k == ignore_index:
            continue
        pk, tk = (pred == k), (tgt == k)
        if not bool(pk.any()) and not bool(tk.any()):
            continue
        pf, tf = pk.float().flatten(1), tk.float().flatten(1)
        d = ((2 * (pf * tf).sum(1) + 1e-5) / ((pf + tf).sum(1) + 1e-5)).mean().item()
        vals[k] = round(d, 4) if rounded else d
    return float(np.nanmean(vals[1:]))


# ---------------------------------------------------------------------------- optimizer groups
def param_groups(named_shapes):
    """trainer.py:812-817: 1-D params and names ending '.bias' get weight_decay 0."""
    decay, no_decay = [], []
    for name, shape in named_shapes:
        (no_decay if (len(shape) == 1 or name.endswith(".bias")) else decay).append(name)
    return decay, no_decay


# ------------------------------------------------------------------------- state-dict geometry
def state_dict_shapes(in_channels, n_cls, n_filters, image_size, transformer_depth):
    """Ordered {key: shape} of the reference model (HDenseFormer.__init__, HDenseFormer.py:178-227;
    SURVEY.md appendix C).  Order = registration order of the reference modules."""
    nf = n_filters
    tok = int(np.prod([d // PATCH for d in image_size]))
    k3 = (3,) * len(image_size)
    k1 = (1,) * len(image_size)
    kp = (PATCH,) * len(image_size)
    dim = 4 * nf
    s = {}
    for m in range(in_channels):
        p = f"attns.{m}"
        s[p + ".position_embeddings"] = (1, tok, dim)
        s[p + ".patch_embeddings.weight"] = (dim, 1) + kp
        s[p + ".patch_embeddings.bias"] = (dim,)
        for b in range(transformer_depth // 4):
            bp = f"{p}.blocks.{b}.0"
            for l in range(LAYERS):
                lp = f"{bp}.layers.{l}"
                s[lp + ".0.weight"] = (GROWTH, dim + GROWTH * l)
                s[lp + ".0.bias"] = (GROWTH,)
                s[lp + ".1.norm.weight"] = (GROWTH,)
                s[lp + ".1.norm.bias"] = (GROWTH,)
                s[lp + ".1.fn.to_qkv.weight"] = (3 * GROWTH, GROWTH)
                s[lp + ".1.fn.to_out.0.weight"] = (GROWTH, GROWTH)
                s[lp + ".1.fn.to_out.0.bias"] = (GROWTH,)
                s[lp + ".2.norm.weight"] = (GROWTH,)
                s[lp + ".2.norm.bias"] = (GROWTH,)
                s[lp + ".2.fn.net.0.weight"] = (2 * GROWTH, GROWTH)
                s[lp + ".2.fn.net.0.bias"] = (2 * GROWTH,)
                s[lp + ".2.fn.net.3.weight"] = (GROWTH, 2 * GROWTH)
                s[lp + ".2.fn.net.3.bias"] = (GROWTH,)
            s[bp + ".out_layer.net.0.weight"] = (2 * GROWTH, dim + LAYERS * GROWTH)
            s[bp + ".out_layer.net.0.bias"] = (2 * GROWTH,)
            s[bp + ".out_layer.net.3.weight"] = (dim, 2 * GROWTH)
            s[bp + ".out_layer.net.3.bias"] = (dim,)

    def upc(name, ci, co):
        s[name + ".double_conv.0.weight"] = (co, ci) + k3
        s[name + ".double_conv.0.bias"] = (co,)

    def basic(name, ci, co):
        s[name + ".conv.weight"] = (co, ci) + k3
        s[name + ".norm.weight"] = (co,)
        s[name + ".norm.bias"] = (co,)

    def convt(name, ci, co):
        s[name + ".weight"] = (ci, co) + k3
        s[name + ".bias"] = (co,)

    def head(name, ci):
        s[name + ".weight"] = (n_cls, ci) + k1
        s[name + ".bias"] = (n_cls,)

    upc("deep_conv", dim * in_channels, 8 * nf)
    upc("up1", 8 * nf, 4 * nf)
    upc("up2", 4 * nf, 2 * nf)
    upc("up3", 2 * nf, nf)
    basic("block_1_1_left", in_channels, nf)
    basic("block_1_2_left", nf, nf)
    basic("block_2_1_left", nf, 2 * nf)
    basic("block_2_2_left", 2 * nf, 2 * nf)
    basic("block_3_1_left", 2 * nf, 4 * nf)
    basic("block_3_2_left", 4 * nf, 4 * nf)
    basic("block_4_1_left", 4 * nf, 8 * nf)
    basic("block_4_2_left", 8 * nf, 8 * nf)
    convt("upconv_3", 8 * nf, 4 * nf)
    basic("block_3_1_right", 8 * nf, 4 * nf)
    basic("block_3_2_right", 4 * nf, 4 * nf)
    convt("upconv_2", 4 * nf, 2 * nf)
    basic("block_2_1_right", 4 * nf, 2 * nf)
    basic("block_2_2_right", 2 * nf, 2 * nf)
    convt("upconv_1", 2 * nf, nf)
    basic("block_1_1_right", 2 * nf, nf)
    basic("block_1_2_right", nf, nf)
    head("conv1x1", nf)
    head("conv1x1_d1", 2 * nf)
    head("conv1x1_d2", 4 * nf)
    head("conv1x1_d3", 8 * nf)
    return s


def det_model(in_channels, n_cls, n_filters, image_size, transformer_depth):
    """Closed-form weights as torch tensors, keyed like the reference state_dict."""
    shapes = state_dict_shapes(in_channels, n_cls, n_filters, image_size, transformer_depth)
    return {k: torch.from_numpy(v) for k, v in detgen.det_state_dict(shapes).items()}


# ---------------------------------------------------------------------------------- train step
class OracleTrainer:
    """The reference's inner training step (trainer.py:369-380) on CPU with torch autograd over the
    functional forward above: forward -> DeepSuper(CEPlusDice) -> backward -> Adam (2 param groups).
    Used as bench.py's cpu_baseline ("port") and by the gradient parity tests."""

    def __init__(self, sd, lr=1e-3, weight_decay=1e-4):
        self.sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        decay, no_decay = param_groups([(k, tuple(v.shape)) for k, v in self.sd.items()])
        self.opt = torch.optim.Adam([
            {"params": [self.sd[k] for k in decay]},
            {"params": [self.sd[k] for k in no_decay], "weight_decay": 0.0}], lr=lr, weight_decay=weight_decay)

    def loss_and_grads(self, x, onehot, drop_seed=None):
        for v in self.sd.values():
            v.grad = None
        outs = forward(x, self.sd, drop_seed)
        loss = deep_super_loss(outs, onehot)
        loss.backward()
        return loss.detach(), [o.detach() for o in outs]

    def step(self, x, onehot, drop_seed=None):
        loss, outs = self.loss_and_grads(x, onehot, drop_seed)
        self.opt.step()
        return loss, outs
