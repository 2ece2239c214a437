"""Pin the oracle (oracle/hdf_oracle.py) against fixtures produced by the REAL reference
(oracle/make_goldens.py, run in the build container).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import detgen
from oracle import hdf_oracle as orc

from conftest import GOLDEN


def _load(name):
    p = os.path.join(GOLDEN, name + ".npz")
    if not os.path.exists(p):
        pytest.skip(f"fixture {name} not generated")
    return np.load(p, allow_pickle=False)


def _sample(t, step):
    sl = (slice(None), slice(None)) + (slice(None, None, step),) * (t.dim() - 2)
    return t.detach()[sl].numpy()


def _rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def _run_model_fixture(name, grads=True):
    g = _load(name)
    in_ch, n_cls, nf, td = [int(v) for v in g["cfg"][:4]]
    size = tuple(int(v) for v in g["cfg"][4:])
    batch = int(g["batch"])
    seed = int(g["train_seed"])
    sd = orc.det_model(in_ch, n_cls, nf, size, td)
    vol = size if len(size) == 3 else (1,) + size             # 2-D configs: depth-1 volumes, squeezed
    x = torch.from_numpy(detgen.det_input(batch, in_ch, vol, tag=name))
    onehot = torch.from_numpy(detgen.one_hot(detgen.det_labels(batch, n_cls, vol, tag=name), n_cls))
    if len(size) == 2:
        x, onehot = x[:, :, 0], onehot[:, :, 0]
    tr = orc.OracleTrainer(sd)
    drop_seed = None if seed < 0 else seed
    outs, inter = orc.forward(x, tr.sd, drop_seed, want_intermediates=True)
    loss = orc.deep_super_loss(outs, onehot)
    # forward: fp32 noise floor of the reference itself is ~3e-6
    for i, o in enumerate(outs):
        step = max(1, int(g["sample_step"]) >> i)
        assert _rel(_sample(o, step), g[f"out{i}"]) < 2e-5, f"out{i}"
    for k, v in inter.items():
        if "inter_" + k not in g.files:          # cat1..3 are exposed for gradient diagnostics only
            continue
        step = int(g["inter_step"]) if v.shape[-1] > 8 else 1
        assert _rel(_sample(v, step), g["inter_" + k]) < 2e-5, k
    assert abs(loss.item() - float(g["loss"])) < 2e-5 * max(1.0, abs(float(g["loss"])))
    assert abs(orc.compute_dice(outs[0].detach(), onehot) - float(g["dice_rounded"])) < 1e-6
    if grads and "grad_norms" in g.files:
        loss.backward()
        names = [str(n) for n in g["grad_names"]]
        assert names == list(tr.sd.keys())          # named_parameters order == state_dict order
        for j, k in enumerate(names):
            gr = tr.sd[k].grad
            ref_n = float(g["grad_norms"][j])
            tol = 5e-4 * max(ref_n, 1e-6) + 2e-7     # dead UpConv biases have ~1e-8 grads
            assert abs(gr.double().norm().item() - ref_n) < tol, (k, gr.norm().item(), ref_n)
        for k in [f[len("gradfull_"):] for f in g.files if f.startswith("gradfull_")]:
            assert _rel(tr.sd[k].grad.numpy(), g["gradfull_" + k]) < 2e-4, k
        ns = g["grad_samples"].shape[1]
        for j, k in enumerate(names):
            gr = tr.sd[k].grad.flatten()
            idx = np.linspace(0, gr.numel() - 1, ns).astype(np.int64)
            scale = float(g["grad_norms"][j]) / np.sqrt(gr.numel()) + 1e-9      # rms entry of this tensor
            assert np.abs(gr[idx].numpy() - g["grad_samples"][j]).max() < 2e-2 * scale + 1e-7, k
        if "param_samples_after" in g.files:
            # ONE Adam step of the reference's own _get_optimizer (trainer.py:793-840) on these gradients
            tr.opt.step()
            for j, k in enumerate(names):
                v = tr.sd[k].detach().flatten()
                idx = np.linspace(0, v.numel() - 1, ns).astype(np.int64)
                # a step is +-lr = 1e-3 per entry; entries whose gradient is at the rounding floor may flip direction
                d = np.abs(v[idx].numpy() - g["param_samples_after"][j])
                ga = np.abs(g["grad_samples"][j])
                live = (ga > 1e-3 * (float(g["grad_norms"][j]) / np.sqrt(v.numel()) + 1e-12)) & (ga > 2e-6)
                assert (d[live] < 2e-5).all(), (k, d.max())


def test_g1_tiny_eval():
    _run_model_fixture("g1_tiny_eval")


def test_g1_tiny_train_hash_dropout():
    """Train mode: the oracle's dropout sites/ordering against the reference run with
    torch.nn.functional.dropout replaced by the same hash masks."""
    _run_model_fixture("g1_tiny_train")


def test_g2_odd_token_grid():
    _run_model_fixture("g2_odd_eval")


def test_g5_full_size_eval():
    _run_model_fixture("g5_full_eval", grads=False)


def test_g4_mid_train_step():
    """nf32 @ 64^3, B=2, train mode (hash dropout), gradients + one Adam step of the reference's optimizer."""
    _run_model_fixture("g4_mid_train")


def test_g6_2d_train_step():
    """HDenseFormer_2D train step (reference models/HDenseFormer_2D.py) incl. gradients and Adam."""
    _run_model_fixture("g6_2d_train")


def test_g5_full_size_train_step():
    """The computation bench.py times (BASELINE configs[1]: 4x128^3, nf32, td24, B=2, train mode): loss, strided
    logits, 1420 gradient norms, 16 gradient samples per tensor and the Adam-updated samples from the real reference.
    ~17 GB and about a minute of CPU."""
    _run_model_fixture("g5_full_train")


@pytest.mark.parametrize("tag", ["c3", "c4", "c4_absent"])
def test_g3_loss(tag):
    g = _load("g3_loss")
    outs = [torch.from_numpy(g[f"{tag}_logits{i}"]).requires_grad_(True) for i in range(4)]
    onehot = torch.from_numpy(g[tag + "_onehot"].astype(np.float32))
    loss = orc.deep_super_loss(outs, onehot)
    loss.backward()
    assert abs(loss.item() - float(g[tag + "_loss"])) < 1e-5
    for i, o in enumerate(outs):
        assert _rel(o.grad.numpy(), g[f"{tag}_grad{i}"]) < 1e-5


@pytest.mark.parametrize("tag", ["deep_w", "cepd_w", "dice_w", "dice_all", "dice_w_all", "ce_w"])
def test_g3w_weighted_loss_forms(tag):
    """class-weighted / ignore_index=None forms (trainer.py:743-771) against the reference's own classes"""
    g = _load("g3w_loss_weighted")
    w = torch.from_numpy(g["class_weight"])
    onehot = torch.from_numpy(g["onehot"].astype(np.float32))
    n = 4 if tag == "deep_w" else 1
    outs = [torch.from_numpy(g[f"logits{i}"]).requires_grad_(True) for i in range(n)]
    loss = {"deep_w": lambda: orc.deep_super_loss(outs, onehot, weight=w),
            "cepd_w": lambda: orc.ce_plus_dice(outs[0], onehot, weight=w),
            "dice_w": lambda: orc.dice_term(outs[0], onehot, weight=w, ignore_index=0),
            "dice_all": lambda: orc.dice_term(outs[0], onehot, ignore_index=None),
            "dice_w_all": lambda: orc.dice_term(outs[0], onehot, weight=w, ignore_index=None),
            "ce_w": lambda: orc.ce_term(outs[0], onehot, weight=w)}[tag]()
    loss.backward()
    assert abs(loss.item() - float(g[tag + "_loss"])) < 1e-5
    for i, o in enumerate(outs):
        assert _rel(o.grad.numpy(), g[f"{tag}_grad{i}"]) < 1e-5


@pytest.mark.parametrize("tag", ["c4", "c4_absent", "c3"])
def test_g7_metric(tag):
    g = _load("g7_metric")
    logits = torch.from_numpy(g[tag + "_logits"])
    onehot = torch.from_numpy(g[tag + "_onehot"].astype(np.float32))
    assert abs(orc.compute_dice(logits, onehot) - float(g[tag + "_dice"])) < 1e-6


def test_g6_2d_plumbing():
    """BASELINE config #1: HDenseFormer_2D 4-ch 256x256 single forward on the CPU path."""
    g = _load("g6_2d")
    sd = orc.det_model(4, 2, 32, (256, 256), 24)
    x = torch.from_numpy(detgen.det_input(1, 4, (1, 256, 256), tag="g6")[:, :, 0])
    with torch.no_grad():
        outs = orc.forward(x, sd)
    for i, o in enumerate(outs):
        assert tuple(o.shape) == tuple(int(v) for v in g[f"shape{i}"])
        assert _rel(_sample(o, 4 if i == 0 else 1), g[f"out{i}"]) < 2e-5


def test_param_groups_rule():
    shapes = orc.state_dict_shapes(2, 3, 16, (32, 32, 32), 8)
    decay, no_decay = orc.param_groups(list(shapes.items()))
    assert "attns.0.position_embeddings" in decay           # 3-D, not '.bias' -> decayed
    assert "block_1_1_left.norm.weight" in no_decay and "conv1x1.bias" in no_decay
    assert len(decay) + len(no_decay) == len(shapes)


def test_state_dict_count_full_config():
    shapes = orc.state_dict_shapes(4, 4, 32, (128, 128, 128), 24)
    assert len(shapes) == 1420
    assert sum(int(np.prod(s)) for s in shapes.values()) == 15_430_000 or \
        abs(sum(int(np.prod(s)) for s in shapes.values()) - 15.43e6) < 0.01e6


# ---------------------------------------------------------------- sliding window, label staging
def test_g8_cal_steps_vs_reference():
    import json
    from oracle import sw_oracle
    g = _load("g8_sliding_window")
    for case in json.loads(str(g["steps_json"])):
        assert sw_oracle.cal_steps(tuple(case["size"]), tuple(case["patch"]), tuple(case["step"])) == case["steps"], case


def test_g8_sliding_window_vs_reference():
    from oracle import sw_oracle
    g = _load("g8_sliding_window")
    in_ch, n_cls, nf, td = [int(v) for v in g["cfg"]]
    patch = tuple(int(v) for v in g["patch"])
    step = tuple(int(v) for v in g["step"])
    size = tuple(int(v) for v in g["image_size"])
    sd = orc.det_model(in_ch, n_cls, nf, patch, td)
    image = detgen.det_input(1, in_ch, size, tag="sw")[0]
    with torch.no_grad():
        lab, mean = sw_oracle.sliding_window(lambda p: orc.forward(p, sd)[0], image, n_cls, patch, step)
    assert _rel(mean[:, ::2, ::2, ::2], g["mean_s2"]) < 1e-4
    assert (lab == g["argmax"]).mean() > 0.9999


def test_g9_to_tensor_onehot_vs_reference():
    from oracle import sw_oracle
    g = _load("g9_to_tensor")
    assert np.array_equal(sw_oracle.to_onehot(g["label"], int(g["n_cls"])), g["onehot"])


def test_g10_normalize_vs_reference():
    """SURVEY 8f-3, second half: MRNormalize / PETandCTNormalize restatement against the reference's own classes."""
    from oracle import sw_oracle
    g = _load("g10_normalize")
    assert np.array_equal(sw_oracle.mr_normalize(g["mr_in"]), g["mr_out"])
    assert np.array_equal(sw_oracle.pet_ct_normalize(g["petct_in"]), g["petct_out"])
    assert np.array_equal(sw_oracle.pet_ct_normalize(g["petct_in"], 40, 400), g["petct_m40_w400_out"])


def test_dropout_hash_mask_statistics():
    """SURVEY 8c 'Dropout': the counter-hash masks (oracle/detgen.py:dropout_keep == csrc/hdf_common.h:hdf_keep) must
    behave like the reference's Bernoulli(0.5) masks: keep probability 0.5 (scale 2.0 keeps the mean), and the masks of
    different sites / seeds / steps (the two ff calls of a layer, to_out, the embedding, consecutive steps) independent."""
    n = 1 << 18
    sites = [detgen.site_id(0, 0, 0, k) for k in range(5)] + [detgen.site_id(1, 2, 3, detgen.KIND_FF2_B),
                                                              detgen.site_id(0, 0, detgen.LAYER_OUT, detgen.KIND_OUT_A),
                                                              detgen.site_emb(0), detgen.site_emb(3)]
    masks = [detgen.dropout_keep(1234, s, n, 0.5).astype(np.float64) for s in sites]
    masks += [detgen.dropout_keep(seed, sites[1], n, 0.5).astype(np.float64) for seed in (1235, 99991, 1234 + 1000003)]
    for m in masks:
        assert abs(m.mean() - 0.5) < 4e-3                      # 4 sigma of a fair coin at n = 2^18 is 3.9e-3
        # no short-range structure along the element index (lag-1 .. lag-32 autocorrelation)
        c = m - m.mean()
        for lag in (1, 2, 7, 32):
            assert abs((c[:-lag] * c[lag:]).mean() / c.var()) < 1e-2
    z = np.stack([m - m.mean() for m in masks])
    corr = (z @ z.T) / n / 0.25
    off = corr - np.diag(np.diag(corr))
    assert np.abs(off).max() < 1e-2, np.abs(off).max()         # pairwise independent sites / seeds / steps
    # other keep probabilities come out right too (threshold arithmetic)
    for p in (0.1, 0.25, 0.9):
        assert abs(detgen.dropout_keep(7, 3, n, p).mean() - (1 - p)) < 4e-3
    # the oracle's Dropper scales kept entries by 1/(1-p)
    x = torch.ones(4, 8, 32)
    y = orc.Dropper(5)(x, sites[0])
    assert set(np.unique(y.numpy()).tolist()) <= {0.0, 2.0} and abs(float(y.mean()) - 1.0) < 0.15
