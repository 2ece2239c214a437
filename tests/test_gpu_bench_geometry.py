"""GPU parity at the geometry the benchmark runs.

tests/test_gpu_model.py checks the plan at 32^3, where the planner never picks the weights-stationary
conv kernel (conv_ws2_kernel, needs >= 48^3), the bf16 stride-1 conv_wgrad2_kernel sees only 32^3 tiles,
the decoder gradient is not split into two dense buffers (needs n_filters % 32 == 0) and the
ConvTranspose3d bias gradients do not come out of the dgrad conv's partial-sum epilogue.  The tests here
run the SAME drop-in surface (models.HDenseFormer / loss.combine_loss / hdf_rt.optim.FlatAdam) at

  * 48^3, n_filters 16 (odd 3^3 token grid)                 fwd+bwd fp32  vs oracle + g2_odd_eval
  * 64^3, n_filters 32, td 8, B=2, TRAIN mode               fwd+bwd+Adam  vs oracle + g4_mid_train (fp32, bf16)
  * 4x128^3, n_filters 32, td 24, B=2, TRAIN mode           fwd+bwd+Adam  vs g5_full_train (fp32), and the exact
    BENCH configuration (bf16 storage) against the fp32 run of the same box.

g4 / g5 fixtures come from the REAL reference run in the build container (oracle/make_goldens.py):
loss, strided logits, 1 420 gradient norms, 16 gradient samples per tensor, a few full gradients and the
parameters after ONE step of the reference's own optimizer (trainer.py:793-840).

Tolerances.  Forward (fp32): logits <= 1e-3 relative (north_star), loss 1e-4.  Gradients (fp32): the reference's
own fp32-vs-fp64 gradient discrepancy is 1.5e-3 .. 6.4e-3 rel-L2 (ReLU / max-pool decisions flip under 1e-6
perturbations), so per-tensor norms must agree to 2e-2 and per-tensor rel-L2 (vs the oracle)
to 2e-2, the last decoder level (not behind any such decision) to 5e-3.  bf16 storage: loss 3e-2 relative,
per-tensor gradient cosine >= 0.95 on every tensor that carries >= 0.1 % of the gradient energy."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import GOLDEN  # noqa: E402
from oracle import detgen  # noqa: E402
from oracle import hdf_oracle as orc  # noqa: E402

DEV = "cuda:0"


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def _rl2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _strided(t, step):
    return t[(slice(None), slice(None)) + (slice(None, None, step),) * 3]


def _fixture(name):
    p = os.path.join(GOLDEN, name + ".npz")
    if not os.path.exists(p):
        pytest.skip(f"fixture {name} not generated")
    g = np.load(p, allow_pickle=False)
    in_ch, n_cls, nf, td = [int(v) for v in g["cfg"][:4]]
    size = tuple(int(v) for v in g["cfg"][4:])
    return g, (in_ch, n_cls, nf, size, td), int(g["batch"]), int(g["train_seed"])


def _build(cfg, dtype):
    from models.HDenseFormer import HDenseFormer
    in_ch, n_cls, nf, size, td = cfg
    net = HDenseFormer(in_ch, n_cls, nf, image_size=size, transformer_depth=td)
    sd = orc.det_model(*cfg)
    net.load_state_dict(sd)
    net = net.to(DEV)
    net.compute_dtype = dtype
    return net, sd


def _data(cfg, batch, tag):
    in_ch, n_cls, nf, size, td = cfg
    x = torch.from_numpy(detgen.det_input(batch, in_ch, size, tag=tag))
    onehot = torch.from_numpy(detgen.one_hot(detgen.det_labels(batch, n_cls, size, tag=tag), n_cls))
    return x, onehot


def _step(net, x, onehot, seed, loss_scale=1.0):
    """forward + DeepSuper(CE+Dice) + backward through the drop-in surface; seed < 0: eval mode.  loss_scale: what
    GradScaler does for float16 storage (trainer.py:374-377); the parameter gradients come back scaled by it."""
    from loss.combine_loss import CEPlusDice, DeepSuperloss
    crit = DeepSuperloss(criterion=CEPlusDice(weight=None, ignore_index=0))
    if seed < 0:
        net.eval()
    else:
        net.train()
        net.set_dropout_seed(seed)          # the masks of the fixture (oracle/detgen.py:dropout_keep)
    for p in net.parameters():
        p.grad = None
    outs = net(x.to(DEV))
    loss = crit(outs, onehot.to(DEV))
    (loss * loss_scale if loss_scale != 1.0 else loss).backward()
    torch.cuda.synchronize()
    return outs, loss


def _check_forward(g, outs, loss, tol_logits, tol_loss):
    for i in range(4):
        e = _rel(_strided(outs[i].detach().float(), max(1, int(g["sample_step"]) >> i)), torch.from_numpy(g[f"out{i}"]))
        print(f"  out{i} rel {e:.3e}")
        assert e < tol_logits, f"out{i}"
    ref = float(g["loss"])
    print(f"  loss {loss.item():.6f} ref {ref:.6f}")
    assert abs(loss.item() - ref) < tol_loss * max(1.0, abs(ref))


def _check_grads_vs_fixture(g, net, norm_tol, sample_tol, tight=()):
    names = [str(n) for n in g["grad_names"]]
    params = dict(net.named_parameters())
    assert names == list(params.keys())
    ns = g["grad_samples"].shape[1]
    total = float(np.sqrt((g["grad_norms"] ** 2).sum()))
    worst_n, worst_s = ("", 0.0), ("", 0.0)
    for j, k in enumerate(names):
        gr = params[k].grad.detach().flatten().cpu()
        ref_n = float(g["grad_norms"][j])
        if ref_n < 1e-6 * total:                      # dead parameters (UpConv conv bias under a non-affine norm)
            assert gr.norm().item() < 1e-5 * total + 1e-6, k
            continue
        en = abs(gr.double().norm().item() - ref_n) / ref_n
        idx = np.linspace(0, gr.numel() - 1, ns).astype(np.int64)
        rms = ref_n / np.sqrt(gr.numel())
        es = float(np.abs(gr[idx].numpy() - g["grad_samples"][j]).max() / (np.abs(g["grad_samples"][j]).max() + rms))
        worst_n = max(worst_n, (k, en), key=lambda kv: kv[1])
        worst_s = max(worst_s, (k, es), key=lambda kv: kv[1])
        assert en < norm_tol, (k, en)
        assert es < sample_tol, (k, es)
    print(f"  grads vs fixture: worst norm {worst_n[0]} {worst_n[1]:.3e}; worst sample {worst_s[0]} {worst_s[1]:.3e}")
    for k in [f[len("gradfull_"):] for f in g.files if f.startswith("gradfull_")]:
        e = _rl2(params[k].grad.detach(), torch.from_numpy(g["gradfull_" + k]))
        print(f"  full grad {k:55s} rel-l2 {e:.3e}")
        assert e < (5e-3 if k in tight else 2e-2), (k, e)


def _check_adam_vs_fixture(g, net, min_live=5000):
    """One optimizer step with the fused flat Adam (lr / weight-decay grouping of trainer.py:793-840) against the
    parameters the reference's own optimizer produced from its gradients."""
    from hdf_rt.optim import FlatAdam
    opt = FlatAdam(net, lr=float(g["adam_lr"]), weight_decay=float(g["adam_wd"]))
    opt.step()
    torch.cuda.synchronize()
    ns = g["grad_samples"].shape[1]
    decay = {k for k, p in net.named_parameters() if not (p.dim() == 1 or k.endswith(".bias"))}
    wd = float(g["adam_wd"])
    bad = 0
    live_n = 0
    for j, (k, p) in enumerate(net.named_parameters()):
        v = p.detach().flatten().cpu()
        idx = np.linspace(0, v.numel() - 1, ns).astype(np.int64)
        d = np.abs(v[idx].numpy() - g["param_samples_after"][j])
        # Adam's first step moves an entry by lr * ge / (|ge| + eps) with ge = g + wd * p on the decayed group: +-lr
        # when |ge| >> eps = 1e-8.  Only its SIGN matters then, so an entry is comparable when the reference's |ge|
        # clears what two correct fp32 gradients may differ by (a fifth of the tensor's rms entry)
        ge = g["grad_samples"][j] + (wd * g["param_samples_before"][j] if k in decay else 0.0)
        live = np.abs(ge) > 0.2 * (float(g["grad_norms"][j]) / np.sqrt(v.numel())) + 2e-6
        bad += int((d[live] > 2e-5).sum())
        live_n += int(live.sum())
    print(f"  adam: {bad} of {live_n} sampled live entries off the reference's updated value")
    assert live_n > min_live and bad <= max(2, live_n // 500)


def _grads_vs_oracle(net, sd, x, onehot, seed, tol, tight=()):
    tr = orc.OracleTrainer(sd)
    ref_loss, ref_outs = tr.loss_and_grads(x, onehot, None if seed < 0 else seed)
    errs = []
    for name, p in net.named_parameters():
        rg = tr.sd[name].grad
        if rg.norm() < 1e-6:
            assert p.grad.norm().item() < 1e-4, name
            continue
        errs.append((name, _rl2(p.grad, rg)))
    errs.sort(key=lambda kv: -kv[1])
    for k, e in errs[:8]:
        print(f"  grad {k:60s} rel-l2={e:.3e}")
    assert errs[0][1] < tol, errs[:4]
    d = dict(errs)
    for name in tight:
        assert d[name] < 5e-3, (name, d[name])
    return tr, ref_loss, ref_outs


def _cosines(net, ref_grads, min_energy=1e-3):
    """per-tensor cosine between net's gradients and ref_grads {name: tensor}; tensors below min_energy of the total
    squared norm are skipped (their direction is rounding noise in any 16-bit run)"""
    tot = sum(float(v.double().norm()) ** 2 for v in ref_grads.values())
    out = []
    for name, p in net.named_parameters():
        b = ref_grads[name].flatten().double().cpu()
        if float(b.norm()) ** 2 < min_energy * tot:
            continue
        a = p.grad.detach().flatten().double().cpu()
        out.append((name, float((a @ b) / (a.norm() * b.norm() + 1e-30))))
    out.sort(key=lambda kv: kv[1])
    return out


def _gate_cosines(cos, low="bf16"):
    """The 16-bit gate on per-tensor gradient cosines (list sorted ascending).  Every tensor >= 0.95 -- except ONE with a
    two-sided band of its own: block_1_2_left.conv.weight, the encoder's second convolution at full resolution, sits at
    0.950 +- 0.002 in every bf16 run (0.9496 at 64^3, 0.9517 at 128^3, 0.9493 at 144^3) and at 0.9927 in float16.
    tools/cos_probe.py (profiles/r06_cos_probe.txt) took it apart: the weight-gradient kernel is exact on its own operands
    (cosine 0.99999997 against a float64 contraction of the tensors the bf16 run stored), its input activation agrees with
    the fp32 run to 0.99999, and all of the loss is in the incoming gradient d(ds_0) (cosine 0.942), which is already at
    0.946 when it arrives from level 1 through the pooling layer: it is the end of the longest backward path of the
    network (bottleneck -> three encoder levels), and on that path the cosine falls level by level (3_1: 0.979, 2_1:
    0.970, 1_2: 0.950) by DECISIONS that flip -- ReLU gates and max-pool arg-maxima recomputed from activations stored
    with an 8-bit mantissa -- not by accumulated rounding: 1 - cos shrinks 6.9x from bf16 to float16, the 8x of the
    rounding step, where rounding noise alone would shrink 64x.  The reference's autocast stores the same tensors in 16
    bits.  A band on both sides pins the value (a floor that follows the measurement is not a gate)."""
    special = "block_1_2_left.conv.weight"
    for name, c in cos:
        if name == special and low == "bf16":
            assert 0.944 <= c <= 0.957, (name, c)
        elif name == special:
            assert c >= 0.985, (name, c)           # float16 storage: 0.9927
        else:
            assert c >= 0.95, (name, c, cos[:4])


def _norm_ratios(net, ref_grads, scale=1.0, min_energy=1e-2):
    """worst |g| / |g_ref| over the tensors that carry >= min_energy of the reference's squared gradient norm"""
    tot = sum(float(v.double().norm()) ** 2 for v in ref_grads.values())
    worst = ("", 1.0)
    for name, p in net.named_parameters():
        b = float(ref_grads[name].double().norm())
        if b * b < min_energy * tot:
            continue
        r = float(p.grad.detach().double().norm()) / scale / b
        if abs(r - 1.0) > abs(worst[1] - 1.0):
            worst = (name, r)
    return worst


# ------------------------------------------------------------------------------------------------- 48^3
def test_backward_fp32_odd_grid_48cubed_vs_oracle_and_golden():
    """g2_odd_eval (48^3, 27 tokens): the fixture already carried the reference's gradient norms / samples; round 1 only
    compared its forward.  48^3 is the smallest volume the weights-stationary conv kernel takes."""
    g, cfg, batch, seed = _fixture("g2_odd_eval")
    net, sd = _build(cfg, "fp32")
    x, onehot = _data(cfg, batch, "g2_odd_eval")
    outs, loss = _step(net, x, onehot, seed)
    _check_forward(g, outs, loss, 1e-3, 1e-4)
    _check_grads_vs_fixture(g, net, 2e-2, 5e-2)
    _grads_vs_oracle(net, sd, x, onehot, seed, 2e-2, tight=("conv1x1.weight", "block_1_2_right.conv.weight",
                                                             "block_1_1_right.conv.weight", "upconv_1.weight",
                                                             "upconv_1.bias"))


# ------------------------------------------------------------------------------------------------- 64^3, nf32, train
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_mid_train_step_vs_reference_golden(dtype):
    g, cfg, batch, seed = _fixture("g4_mid_train")
    net, sd = _build(cfg, dtype)
    x, onehot = _data(cfg, batch, "g4_mid_train")
    outs, loss = _step(net, x, onehot, seed)
    if dtype == "fp32":
        _check_forward(g, outs, loss, 1e-3, 1e-4)
        _check_grads_vs_fixture(g, net, 2e-2, 5e-2, tight=("conv1x1.weight", "upconv_1.bias", "block_1_1_right.norm.weight"))
        _grads_vs_oracle(net, sd, x, onehot, seed, 2e-2,
                         tight=("conv1x1.weight", "block_1_2_right.conv.weight", "block_1_1_right.conv.weight",
                                "upconv_1.weight", "upconv_1.bias"))
        _check_adam_vs_fixture(g, net)
    else:
        assert outs[0].dtype == torch.bfloat16
        ref = float(g["loss"])
        assert abs(loss.item() - ref) < 3e-2 * abs(ref), (loss.item(), ref)
        tr = orc.OracleTrainer(sd)
        tr.loss_and_grads(x, onehot, seed)
        cos = _cosines(net, {k: v.grad for k, v in tr.sd.items()})
        for k, c in cos[:6]:
            print(f"  bf16 cosine {k:60s} {c:.4f}")
        _gate_cosines(cos)


# ------------------------------------------------------------------------------------------------- BENCH shape
def test_full_size_train_step_fp32_and_bench_bf16_vs_reference_golden():
    """BASELINE configs[1] exactly as bench.py runs it: 4x128^3, n_filters 32, transformer_depth 24, batch 2, train
    mode (dropout on), forward + DeepSuper(CE+Dice) + backward + Adam.  fp32 storage path against the real
    reference's fixture; then the bf16 storage path (what BENCH times) against that fp32 run, tensor by tensor."""
    g, cfg, batch, seed = _fixture("g5_full_train")
    x, onehot = _data(cfg, batch, "g5_full_train")
    net, _ = _build(cfg, "fp32")
    outs, loss = _step(net, x, onehot, seed)
    _check_forward(g, outs, loss, 1e-3, 1e-4)
    _check_grads_vs_fixture(g, net, 2e-2, 5e-2, tight=("conv1x1.weight", "upconv_1.bias", "block_1_1_right.norm.weight"))
    ref_grads = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
    ref_outs = [o.detach().float().clone() for o in outs]
    _check_adam_vs_fixture(g, net)
    del net, outs
    torch.cuda.empty_cache()

    net, _ = _build(cfg, "bf16")
    outs, loss = _step(net, x, onehot, seed)
    assert outs[0].dtype == torch.bfloat16
    ref = float(g["loss"])
    print(f"  bf16 loss {loss.item():.5f} ref {ref:.5f}")
    assert abs(loss.item() - ref) < 3e-2 * abs(ref)
    for i in range(4):
        e = _rl2(outs[i].detach().float(), ref_outs[i])
        print(f"  bf16 out{i} rel-l2 vs fp32 {e:.3e}")
        assert e < 3e-2
        # ... and against the REAL reference's strided logits of the fixture (not only against this box's fp32 run)
        ef = _rl2(_strided(outs[i].detach().float(), max(1, int(g["sample_step"]) >> i)), torch.from_numpy(g[f"out{i}"]))
        print(f"  bf16 out{i} rel-l2 vs fixture {ef:.3e}")
        assert ef < 3e-2
    agree = (outs[0].detach().float().argmax(1) == ref_outs[0].argmax(1)).float().mean().item()
    print("  bf16 argmax agreement", agree)
    assert agree >= 0.985
    cos = _cosines(net, ref_grads)
    for k, c in cos[:8]:
        print(f"  bf16 cosine {k:60s} {c:.4f}")
    _gate_cosines(cos)
    mine = torch.cat([p.grad.flatten() for p in net.parameters()]).double()
    theirs = torch.cat([ref_grads[k].flatten() for k, _ in net.named_parameters()]).double()
    whole = float((mine @ theirs) / (mine.norm() * theirs.norm()))
    print("  bf16 whole-gradient cosine", whole)
    assert whole > 0.98
    # a bf16-only defect of a few per cent in one layer moves that layer's gradient norm: every tensor with >= 1 % of the
    # gradient energy must keep its norm within 5 % of the fp32 run's (which the fixture pinned above), and its norm
    # within 6 % of the reference's own
    worst = _norm_ratios(net, ref_grads)
    print(f"  bf16 worst gradient-norm ratio vs fp32 {worst[0]} {worst[1]:.4f}")
    assert abs(worst[1] - 1.0) < 0.05, worst
    names = [str(n) for n in g["grad_names"]]
    fix = {k: float(v) for k, v in zip(names, g["grad_norms"])}
    tot = sum(v * v for v in fix.values())
    for k, p in net.named_parameters():
        if fix[k] ** 2 >= 1e-2 * tot:
            r = float(p.grad.detach().double().norm()) / fix[k]
            assert abs(r - 1.0) < 0.06, (k, r)


# ------------------------------------------------------------------------------------------------- BASELINE configs[3], [4]
@pytest.mark.parametrize("name,cfg,low", [
    ("hecktor_144", (2, 3, 32, (144, 144, 144), 24), "bf16"),      # configs[3]: PET/CT 2-modal, 3-class, N = 729 tokens
    ("nf48_160", (4, 4, 48, (160, 160, 160), 24), "fp16"),         # configs[4]: n_filters 48, N = 1000, fp16
])
def test_baseline_config_shapes_forward_vs_oracle_and_low_precision_step(name, cfg, low):
    """fp32 eval forward against the oracle run on this box's CPU (strided logits <= 1e-3, forward Dice <= 1e-4), then ONE
    train step (dropout on) in fp32 whose loss, logits and EVERY parameter gradient are compared with the oracle's
    train step on the same inputs and dropout masks (tensor by tensor, rel-L2 <= 2e-2, the reference's own fp32 noise
    floor), and the same step in the config's 16-bit storage type: finite loss within 3e-2, per-tensor gradient cosine
    >= 0.95 and gradient-norm ratio within 10 % against that oracle-checked fp32 step."""
    from hdf_rt.loss_fn import compute_dice
    batch = 1
    x, onehot = _data(cfg, batch, name)
    net, sd = _build(cfg, "fp32")
    net.eval()
    with torch.no_grad():
        outs = net(x.to(DEV))
        ref = orc.forward(x, sd)
    for i in range(4):
        e = _rel(outs[i], ref[i])
        print(f"  {name} out{i} rel {e:.3e}")
        assert e < 1e-3
    d_hip = compute_dice(outs[0], onehot.to(DEV))
    d_ref = orc.compute_dice(ref[0], onehot)
    print(f"  {name} dice {float(d_hip):.4f} ref {d_ref:.4f}")
    assert abs(float(d_hip) - d_ref) <= 1e-4
    del ref
    outs, loss32 = _step(net, x, onehot, 777)
    ref_grads = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
    # the fp32 backward of THIS plan (96 / 192-byte channel rows, 9^3 / 10^3 token grids) against the oracle at full size
    tr, ref_loss, ref_outs = _grads_vs_oracle(net, sd, x, onehot, 777, 2e-2,
                                              tight=("conv1x1.weight", "block_1_2_right.conv.weight"))
    print(f"  {name} train loss {loss32.item():.6f} oracle {ref_loss.item():.6f}")
    assert abs(loss32.item() - ref_loss.item()) < 1e-4 * max(1.0, abs(ref_loss.item()))
    for i in range(4):
        assert _rel(outs[i].detach().float(), ref_outs[i]) < 1e-3, f"train out{i}"
    del net, outs, tr, ref_outs
    torch.cuda.empty_cache()
    net, _ = _build(cfg, low)
    # float16 needs the reference's loss scaling: d loss / d logits is ~1/voxels = 2.4e-7 at 160^3, below the smallest
    # normal half (6.1e-5); GradScaler's default initial scale is 65536 (cosines are scale-invariant)
    outs, loss = _step(net, x, onehot, 777, loss_scale=65536.0 if low == "fp16" else 1.0)
    assert outs[0].dtype == (torch.bfloat16 if low == "bf16" else torch.float16)
    print(f"  {name} {low} loss {loss.item():.5f} fp32 {loss32.item():.5f}")
    assert np.isfinite(loss.item()) and abs(loss.item() - loss32.item()) < 3e-2 * abs(loss32.item())
    cos = _cosines(net, ref_grads)
    for k, c in cos[:5]:
        print(f"  {low} cosine {k:60s} {c:.4f}")
    _gate_cosines(cos, low)
    assert all(bool(torch.isfinite(p.grad).all()) for p in net.parameters())
    scale = 65536.0 if low == "fp16" else 1.0
    worst = _norm_ratios(net, ref_grads, scale)
    print(f"  {low} worst gradient-norm ratio {worst[0]} {worst[1]:.4f}")
    assert abs(worst[1] - 1.0) < 0.10, worst


def test_weight_gradients_on_the_side_stream_are_reproducible():
    """The conv / transposed-conv weight gradients run on the plan's side stream next to the rest of backward
    (csrc/plan.hip Exec).  They are summed in a fixed order, so the same step run three times must give bitwise equal
    gradients for every conv weight: a weight gradient that reads a dy buffer the main stream has already overwritten, or
    a join that comes too early, shows up as a difference.  (Biases, LayerNorm parameters and the heads use float
    atomics and are compared with a tolerance.)  bf16 storage at 64^3 x 2: every kernel family of the benchmark."""
    cfg, batch, tag = (4, 4, 32, (64, 64, 64), 8), 2, "g4_mid_train"
    net, _ = _build(cfg, "bf16")
    x, onehot = _data(cfg, batch, tag)
    runs = []
    for _ in range(3):
        _step(net, x, onehot, 4321)
        runs.append({n: p.grad.detach().clone() for n, p in net.named_parameters()})
    exact = [n for n in runs[0] if n.endswith("conv.weight") or (n.startswith("upconv_") and n.endswith(".weight")) or
             n.endswith("double_conv.0.weight")]
    assert len(exact) >= 20
    for n in exact:
        assert torch.equal(runs[0][n], runs[1][n]) and torch.equal(runs[0][n], runs[2][n]), n
    for n in runs[0]:
        if runs[0][n].norm() > 0:
            assert _rl2(runs[1][n], runs[0][n]) < 1e-4, n


def test_probe_events_bracket_the_dominant_conv_launch_and_change_nothing():
    """hdf_plan_set_probe (include/hdf.h): two caller-owned events recorded around the launch of block_1_1_right's conv in
    every following forward -- what bench.py's roofline.in_step is measured with.  The logits must be bit-identical with
    and without the probe, the bracketed time positive and a small part of the forward, and NULL, NULL switches it off."""
    from hdf_rt._lib import check, lib
    cfg, batch, tag = (4, 4, 32, (64, 64, 64), 8), 2, "g4_mid_train"
    net, _ = _build(cfg, "bf16")
    net.eval()
    x, _ = _data(cfg, batch, tag)
    with torch.no_grad():
        ref = [o.clone() for o in net(x.to(DEV))]
        rt = net._last_rt
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(), e1.record()
        torch.cuda.synchronize()
        check(lib().hdf_plan_set_probe(rt.plan.h, e0.cuda_event, e1.cuda_event), "set_probe")
        f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        f0.record()
        got = net(x.to(DEV))
        f1.record()
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(ref, got))
        t_conv, t_fwd = e0.elapsed_time(e1), f0.elapsed_time(f1)
        assert 0.0 < t_conv < 0.5 * t_fwd, (t_conv, t_fwd)
        check(lib().hdf_plan_set_probe(rt.plan.h, None, None), "set_probe")
        assert lib().hdf_plan_set_probe(rt.plan.h, e0.cuda_event, None) != 0      # one event without the other: refused
        got2 = net(x.to(DEV))
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(ref, got2))
