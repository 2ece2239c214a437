"""Deterministic, closed-form generators shared by the ORACLE side (test infrastructure).

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import anything under oracle/.  The product package never does.

Two things live here:

* `mix32` / `dropout_keep` -- the counter-based hash that decides dropout keep-masks.  The HIP
  kernels (h-denseformer_amd/csrc/hdf_common.h: hdf_mix32 / hdf_keep) implement the same integer
  recipe, so train-mode (dropout ON) runs of the HIP path and of the oracle produce identical
  masks and can be compared element-wise.  The reference's dropout sites are
  /root/reference/models/HDenseFormer.py:39,41 (DenseForward), :61 (to_out), :130,138 (embedding).

* `det_tensor` / `det_state_dict` -- closed-form weights keyed by (state_dict key, flat index), so
  the GPU box can regenerate exactly the weights the golden fixtures were made with, without the
  reference being present.
"""
import zlib

import numpy as np

U32 = np.uint32


def mix32(x):
    """lowbias32 integer finaliser on uint32 arrays (wrap-around arithmetic)."""
    x = np.asarray(x, dtype=np.uint32).copy()
    with np.errstate(over="ignore"):
        x ^= x >> U32(16)
        x *= U32(0x7FEB352D)
        x ^= x >> U32(15)
        x *= U32(0x846CA68B)
        x ^= x >> U32(16)
    return x


# dropout site ids -----------------------------------------------------------------------------
KIND_ATTN_OUT, KIND_FF1_A, KIND_FF1_B, KIND_FF2_A, KIND_FF2_B = 0, 1, 2, 3, 4
KIND_OUT_A, KIND_OUT_B = 0, 1          # used with layer index 4 (block-level out_layer)
LAYER_OUT = 4
SITE_EMB_B, SITE_EMB_L, SITE_EMB_K = 63, 7, 7


def site_id(m, b, l, kind):
    """modality m, block b, dense layer l (4 = out_layer), kind -> integer site id."""
    return ((m * 64 + b) * 8 + l) * 8 + kind


def site_emb(m):
    return site_id(m, SITE_EMB_B, SITE_EMB_L, SITE_EMB_K)


def dropout_keep(seed, site, n_elems, p):
    """Boolean keep-mask of length n_elems for (seed, site); element index = flat row-major index
    of the [B, N, width] activation the dropout is applied to."""
    with np.errstate(over="ignore"):
        k0 = mix32(U32(seed & 0xFFFFFFFF) ^ (U32(site) * U32(0x9E3779B1)))
        idx = np.arange(n_elems, dtype=np.uint32)
        h = mix32(idx + k0)
    thresh = int(round((1.0 - p) * (1 << 24)))
    return (h >> U32(8)) < U32(thresh)


# closed-form weights ----------------------------------------------------------------------------
def _uniform(key, n):
    """n floats in [-1, 1), a pure function of (key string, index)."""
    with np.errstate(over="ignore"):
        k0 = mix32(U32(zlib.crc32(key.encode()) & 0xFFFFFFFF))
        h = mix32(np.arange(n, dtype=np.uint32) * U32(0x9E3779B1) + k0)
    return (h >> U32(8)).astype(np.float64) / float(1 << 23) - 1.0


def det_tensor(key, shape):
    """Closed-form fp32 tensor for a state_dict entry.  Magnitudes follow the usual fan-in rule so
    activations stay O(1) through the network; norm weights sit around 1, biases are small, and the
    (zero-initialised in the reference) position embeddings get small non-zero values so that the
    addition is actually exercised."""
    n = int(np.prod(shape))
    u = _uniform(key, n)
    if key.endswith("position_embeddings"):
        v = 0.1 * u
    elif key.endswith("norm.weight"):
        v = 1.0 + 0.25 * u
    elif key.endswith("norm.bias"):
        v = 0.1 * u
    elif key.endswith(".bias"):
        v = 0.05 * u
    else:
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else int(shape[0])
        if "upconv_" in key and len(shape) == 5:      # ConvTranspose3d weight [Cin, Cout, 3,3,3]
            fan_in = int(shape[0]) * 27 // 8 + 1      # ~27/8 taps hit each output voxel
        v = u * np.sqrt(3.0 / fan_in)
    return v.astype(np.float32).reshape(shape)


def det_state_dict(shapes):
    """shapes: ordered mapping key -> shape.  Returns dict key -> np.float32 array."""
    return {k: det_tensor(k, tuple(s)) for k, s in shapes.items()}


def det_input(batch, channels, size, tag="x"):
    """Synthetic volume in [0,1) (the range after MRNormalize, data_loader.py:39-50) and an integer
    label map, both closed-form."""
    d, h, w = size
    x = (_uniform(tag + ".image", batch * channels * d * h * w) * 0.5 + 0.5).astype(np.float32)
    return x.reshape(batch, channels, d, h, w)


def det_labels(batch, n_cls, size, tag="x"):
    d, h, w = size
    u = _uniform(tag + ".label", batch * d * h * w) * 0.5 + 0.5
    # blocky labels: quantise a smooth-ish field so that classes form regions, not salt and pepper
    zz, yy, xx = np.meshgrid(np.arange(d), np.arange(h), np.arange(w), indexing="ij")
    field = (np.sin(zz / 5.0)[None] + np.cos(yy / 7.0)[None] + np.sin(xx / 3.0 + 1.0)[None]) / 6.0 + 0.5
    field = field + 0.15 * (u.reshape(batch, d, h, w) - 0.5) + 0.07 * np.arange(batch)[:, None, None, None]
    lab = np.clip((field * n_cls).astype(np.int64), 0, n_cls - 1)
    return lab


def one_hot(lab, n_cls):
    """[B,D,H,W] int -> fp32 one-hot [B,n_cls,D,H,W] (data_loader.py:146-151 format)."""
    oh = (lab[:, None] == np.arange(n_cls)[None, :, None, None, None]).astype(np.float32)
    return oh
