"""GPU parity of the whole hot path through the drop-in surface (models.HDenseFormer / loss.combine_loss)
against the oracle (oracle/hdf_oracle.py, pinned to the real reference by tests/golden) and against
the golden fixtures themselves.

Tolerances (BASELINE.json north_star): fp32 path logits <= 1e-3 relative (max-abs / max-ref; measured
~1e-5), forward Dice within 1e-4; bf16 storage path rel-L2 <= 3e-2 and argmax agreement >= 98.5 %
(what the reference's own bf16 autocast reaches)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import GOLDEN  # noqa: E402
from oracle import detgen  # noqa: E402
from oracle import hdf_oracle as orc  # noqa: E402

DEV = "cuda:0"
CFG_TINY = (2, 3, 16, (32, 32, 32), 8)
CFG_ODD = (2, 2, 16, (48, 48, 48), 4)


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def _rl2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _strided(t, step):
    return t[(slice(None), slice(None)) + (slice(None, None, step),) * 3]


def _build(cfg, dtype=None):
    from models.HDenseFormer import HDenseFormer
    in_ch, n_cls, nf, size, td = cfg
    net = HDenseFormer(in_ch, n_cls, nf, image_size=size, transformer_depth=td)
    sd = orc.det_model(*cfg)
    assert list(net.state_dict().keys()) == list(sd.keys())
    net.load_state_dict(sd)
    net = net.to(DEV)
    net.compute_dtype = dtype
    return net, sd


def _data(cfg, batch, tag):
    in_ch, n_cls, nf, size, td = cfg
    x = torch.from_numpy(detgen.det_input(batch, in_ch, size, tag=tag))
    onehot = torch.from_numpy(detgen.one_hot(detgen.det_labels(batch, n_cls, size, tag=tag), n_cls))
    return x, onehot


def _report(name, pairs):
    worst = max(pairs, key=lambda kv: kv[1])
    print(f"[{name}] worst: {worst[0]} = {worst[1]:.3e}")
    return worst


@pytest.mark.parametrize("cfg,batch,tag", [(CFG_TINY, 2, "g1_tiny_eval"), (CFG_ODD, 1, "g2_odd_eval")])
def test_forward_fp32_vs_oracle_and_golden(cfg, batch, tag):
    net, sd = _build(cfg)
    net.eval()
    x, onehot = _data(cfg, batch, tag)
    with torch.no_grad():
        outs = net(x.to(DEV))
        ref_outs, inter = orc.forward(x, sd, None, want_intermediates=True)
    torch.cuda.synchronize()
    rt = net._last_rt
    errs = []
    for k, v in inter.items():
        if k.startswith(("dec", "cat")) or k == "at3":   # never materialised on the HIP path (fused into consumers)
            continue
        errs.append((k, _rel(rt.read_buffer(k), v)))
    for i in range(4):
        errs.append((f"out{i}", _rel(outs[i], ref_outs[i])))
    for k, e in errs:
        print(f"  {k:12s} rel={e:.3e}")
    assert _report(tag, errs)[1] < 1e-3
    # golden from the REAL reference
    g = np.load(os.path.join(GOLDEN, tag + ".npz"))
    for i in range(4):
        assert _rel(_strided(outs[i], max(1, int(g["sample_step"]) >> i)), torch.from_numpy(g[f"out{i}"])) < 1e-3
    # forward Dice within 1e-4 of the reference's
    from hdf_rt.loss_fn import compute_dice
    assert abs(compute_dice(outs[0], onehot.to(DEV)) - float(g["dice_rounded"])) <= 1e-4


CFG_SIX = (3, 6, 16, (32, 32, 32), 4)     # six classes: the 8-class-slot forms of the head and loss kernels


@pytest.mark.parametrize("train,cfg", [(False, CFG_TINY), (True, CFG_TINY), (True, CFG_SIX)])
def test_backward_fp32_vs_oracle(train, cfg):
    batch, tag = 2, "g1_tiny_eval"
    from loss.combine_loss import CEPlusDice, DeepSuperloss
    net, sd = _build(cfg)
    net.train(train)
    x, onehot = _data(cfg, batch, tag)
    seed = None
    if train:
        seed = (net.dropout_seed * 1000003 + net._step + 1) & 0xFFFFFFFF
    crit = DeepSuperloss(criterion=CEPlusDice(weight=None, ignore_index=0))
    outs = net(x.to(DEV))
    loss = crit(outs, onehot.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    # the oracle runs in float64 here: on this fixture its own fp32 gradients are up to 1.2e-2 rel-L2 away from its
    # float64 ones (attns.0.blocks.0.0.layers.0.1.norm.*), i.e. an fp32 oracle is the less accurate side of the comparison
    tr = orc.OracleTrainer({k: v.double() for k, v in sd.items()})
    ref_loss, ref_outs = tr.loss_and_grads(x.double(), onehot.double(), seed)
    print("loss", loss.item(), ref_loss.item())
    for i in range(4):
        assert _rel(outs[i].detach(), ref_outs[i]) < 1e-3, f"out{i}"
    assert abs(loss.item() - ref_loss.item()) < 1e-4 * max(1.0, abs(ref_loss.item()))
    errs = []
    for name, p in net.named_parameters():
        rg = tr.sd[name].grad
        if rg.norm() < 1e-6:                       # dead parameters (UpConv conv bias under non-affine IN)
            assert p.grad.norm().item() < 1e-4, name
            continue
        errs.append((name, _rl2(p.grad, rg)))
    errs.sort(key=lambda kv: -kv[1])
    for k, e in errs[:12]:
        print(f"  grad {k:60s} rel-l2={e:.3e}")
    # Gradient tolerance: ReLU / max-pool decisions flip under 1e-6 perturbations on this fixture (the reference
    # path's own fp32-vs-fp64 gradient discrepancy is 1.5e-3 .. 1.2e-2 rel-L2, DESIGN.md "parity"), so an fp32
    # implementation agrees with the float64 oracle to ~1e-2 at best.
    # The last decoder level does not sit behind those decisions and must agree tightly.
    assert errs[0][1] < 1e-2
    d = dict(errs)
    for name in ("conv1x1.weight", "block_1_2_right.conv.weight", "block_1_1_right.conv.weight", "upconv_1.weight"):
        assert d[name] < 5e-3, (name, d[name])


def test_forward_bf16_storage():
    cfg, batch, tag = CFG_TINY, 2, "g1_tiny_eval"
    net, sd = _build(cfg, "bf16")
    net.eval()
    x, _ = _data(cfg, batch, tag)
    with torch.no_grad():
        outs = net(x.to(DEV))
        ref = orc.forward(x, sd)
    assert outs[0].dtype == torch.bfloat16
    for i in range(4):
        print(f"  bf16 out{i} rel-l2 {_rl2(outs[i].float(), ref[i]):.3e}")
        assert _rl2(outs[i].float(), ref[i]) < 3e-2
    agree = (outs[0].float().argmax(1).cpu() == ref[0].argmax(1)).float().mean().item()
    print("  argmax agreement", agree)
    assert agree >= 0.985


def test_autocast_selects_bf16_and_backward_runs():
    from loss.combine_loss import CEPlusDice, DeepSuperloss
    cfg, batch, tag = CFG_TINY, 2, "g1_tiny_eval"
    net, sd = _build(cfg)
    net.train(False)
    x, onehot = _data(cfg, batch, tag)
    crit = DeepSuperloss(criterion=CEPlusDice(weight=None, ignore_index=0))
    with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
        outs = net(x.to(DEV))
    loss = crit(outs, onehot.to(DEV))          # outside autocast, like trainer.py:371
    loss.backward()
    assert outs[0].dtype == torch.bfloat16
    tr = orc.OracleTrainer(sd)
    ref_loss, _ = tr.loss_and_grads(x, onehot, None)
    assert abs(loss.item() - ref_loss.item()) < 3e-2 * abs(ref_loss.item())
    # bf16 storage: gradient direction must agree with the fp32 reference
    mine = torch.cat([p.grad.flatten().cpu() for _, p in net.named_parameters()]).double()
    theirs = torch.cat([tr.sd[n].grad.flatten() for n, _ in net.named_parameters()]).double()
    cos = float((mine @ theirs) / (mine.norm() * theirs.norm()))
    print("  bf16 whole-gradient cosine", cos)
    assert cos > 0.98
    for name in ("block_1_2_right.conv.weight", "block_2_1_left.conv.weight", "upconv_2.weight", "deep_conv.double_conv.0.weight"):
        a, b = dict(net.named_parameters())[name].grad.flatten().cpu().double(), tr.sd[name].grad.flatten().double()
        c = float((a @ b) / (a.norm() * b.norm()))
        print(f"  bf16 cosine {name}: {c:.4f}")
        assert c > 0.95, name


@pytest.mark.parametrize("tag", ["c3", "c4", "c4_absent"])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_fused_loss_vs_reference_golden(tag, dt):
    from loss.combine_loss import CEPlusDice, DeepSuperloss
    g = np.load(os.path.join(GOLDEN, "g3_loss.npz"))
    outs = [torch.from_numpy(g[f"{tag}_logits{i}"]).to(DEV).to(dt).requires_grad_(True) for i in range(4)]
    onehot = torch.from_numpy(g[tag + "_onehot"].astype(np.float32)).to(DEV)
    crit = DeepSuperloss(criterion=CEPlusDice(weight=None, ignore_index=0))
    loss = crit(outs, onehot)
    loss.backward()
    tol = 2e-5 if dt == torch.float32 else 2e-2
    assert abs(loss.item() - float(g[tag + "_loss"])) < tol * 10
    for i, o in enumerate(outs):
        assert _rel(o.grad.float(), torch.from_numpy(g[f"{tag}_grad{i}"])) < tol * 5


@pytest.mark.parametrize("tag", ["deep_w", "cepd_w", "dice_w", "dice_all", "dice_w_all", "ce_w"])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_weighted_loss_forms_vs_reference_golden(tag, dt):
    """What trainer.py:743-771 builds with a class_weight list (and DiceLoss(ignore_index=None)), through the drop-in
    loss modules, against fixtures from the reference's own classes (oracle/make_goldens.py --only g3w)."""
    from loss.combine_loss import CEPlusDice, DeepSuperloss
    from loss.cross_entropy import CrossentropyLoss
    from loss.dice_loss import DiceLoss
    g = np.load(os.path.join(GOLDEN, "g3w_loss_weighted.npz"))
    w = torch.tensor(g["class_weight"].tolist())          # a CPU tensor, as trainer.py:745 makes it
    onehot = torch.from_numpy(g["onehot"].astype(np.float32)).to(DEV)
    n = 4 if tag == "deep_w" else 1
    outs = [torch.from_numpy(g[f"logits{i}"]).to(DEV).to(dt).requires_grad_(True) for i in range(n)]
    crit = {"deep_w": DeepSuperloss(criterion=CEPlusDice(weight=w, ignore_index=0)),
            "cepd_w": CEPlusDice(weight=w, ignore_index=0),
            "dice_w": DiceLoss(weight=w, ignore_index=0, p=1),
            "dice_all": DiceLoss(weight=None, ignore_index=None),
            "dice_w_all": DiceLoss(weight=w, ignore_index=None),
            "ce_w": CrossentropyLoss(weight=w)}[tag]
    loss = crit(outs, onehot) if n > 1 else crit(outs[0], onehot)
    loss.backward()
    tol = 2e-5 if dt == torch.float32 else 2e-2
    assert abs(loss.item() - float(g[tag + "_loss"])) < tol * 10
    for i, o in enumerate(outs):
        assert _rel(o.grad.float(), torch.from_numpy(g[f"{tag}_grad{i}"])) < tol * 5


@pytest.mark.parametrize("shape", [(8, 16, 24), (16, 8, 40), (24, 40)])
@pytest.mark.parametrize("weighted", [False, True])
@pytest.mark.parametrize("c", [4, 6])
def test_fused_loss_on_rows_that_are_not_multiples_of_four_vs_oracle(shape, weighted, c):
    """The one-launch loss kernels take 4 voxels of a row per thread where the row width of a scale allows it and one
    otherwise (widths 24/12/6/3, 40/20/10/5; 2-D logits too), with 4 or 8 class slots in registers: loss and logit
    gradients against the oracle in fp32."""
    from loss.combine_loss import CEPlusDice, DeepSuperloss
    gen = torch.Generator().manual_seed(11)
    n = 2
    lab = torch.randint(0, c, (n,) + shape, generator=gen)
    onehot = torch.nn.functional.one_hot(lab, c).movedim(-1, 1).float().contiguous()
    outs = [torch.randn((n, c) + tuple(d >> i for d in shape), generator=gen) for i in range(4)]
    w = torch.tensor([0.2, 1.0, 2.0, 0.5, 1.5, 0.7][:c]) if weighted else None
    ref_in = [o.double().requires_grad_(True) for o in outs]
    ref = orc.deep_super_loss(ref_in, onehot.double(), weight=None if w is None else w.double(), ignore_index=0)
    ref.backward()
    mine_in = [o.to(DEV).requires_grad_(True) for o in outs]
    loss = DeepSuperloss(criterion=CEPlusDice(weight=w, ignore_index=0))(mine_in, onehot.to(DEV))
    loss.backward()
    assert abs(loss.item() - ref.item()) < 2e-5 * abs(ref.item())
    for a, b in zip(mine_in, ref_in):
        assert _rel(a.grad, b.grad) < 1e-4


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_fused_loss_on_views_that_are_not_16_byte_aligned(dt):
    """The 4-voxel loads of the loss kernels need aligned rows; logits handed over as a view one element into its
    storage (contiguous, but 2 or 4 bytes off) take the one-voxel form: same loss, same gradients."""
    from loss.combine_loss import CEPlusDice, DeepSuperloss
    g = np.load(os.path.join(GOLDEN, "g3_loss.npz"))
    onehot = torch.from_numpy(g["c4_onehot"].astype(np.float32)).to(DEV)
    crit = DeepSuperloss(criterion=CEPlusDice(weight=None, ignore_index=0))

    def run(shift):
        outs = []
        for i in range(4):
            ref = torch.from_numpy(g[f"c4_logits{i}"]).to(DEV).to(dt)
            buf = torch.zeros(ref.numel() + 8, device=DEV, dtype=dt)
            v = buf[shift:shift + ref.numel()].view(ref.shape)
            v.copy_(ref)
            assert v.is_contiguous() and (v.data_ptr() % 16 != 0) == (shift % (16 // v.element_size()) != 0)
            outs.append(v.requires_grad_(True))
        loss = crit(outs, onehot)
        loss.backward()
        return loss.item(), [o.grad.float() for o in outs]

    l0, g0 = run(0)
    l1, g1 = run(1)
    assert abs(l0 - l1) < 1e-5 * abs(l0)
    for a, b in zip(g0, g1):
        assert _rel(b, a) < (1e-5 if dt == torch.float32 else 1e-2)


@pytest.mark.parametrize("tag", ["c4", "c4_absent", "c3"])
def test_dice_metric_vs_reference_golden(tag):
    from hdf_rt.loss_fn import compute_dice
    g = np.load(os.path.join(GOLDEN, "g7_metric.npz"))
    logits = torch.from_numpy(g[tag + "_logits"]).to(DEV)
    onehot = torch.from_numpy(g[tag + "_onehot"].astype(np.float32)).to(DEV)
    d = compute_dice(logits, onehot)
    assert abs(d - float(g[tag + "_dice"])) < 1e-6
    assert abs(d.item() - float(g[tag + "_dice"])) < 1e-6      # trainer.py:391 calls dice.item()


def test_flat_adam_matches_torch_adam():
    from hdf_rt.optim import FlatAdam
    cfg = CFG_TINY
    net, sd = _build(cfg)
    ref = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    decay, no_decay = orc.param_groups([(k, tuple(v.shape)) for k, v in ref.items()])
    topt = torch.optim.Adam([{"params": [ref[k] for k in decay]},
                             {"params": [ref[k] for k in no_decay], "weight_decay": 0.0}], lr=1e-3, weight_decay=1e-4)
    opt = FlatAdam(net, lr=1e-3, weight_decay=1e-4)
    gflat = net.flat_grads()
    for step in range(3):
        g = torch.Generator().manual_seed(step)
        for (name, p), v in zip(net.named_parameters(), net._grad_views):
            gr = torch.randn(p.shape, generator=g) * 0.01
            v.copy_(gr.to(DEV))
            ref[name].grad = gr
        opt.step()
        topt.step()
    assert gflat is net.flat_grads()
    worst = max(_rel(p.detach(), ref[name].detach()) for name, p in net.named_parameters())
    print("  adam worst rel", worst)
    assert worst < 1e-5


def test_train_step_changes_params_and_state_dict_roundtrip():
    from hdf_rt.optim import FlatAdam
    from loss.combine_loss import CEPlusDice, DeepSuperloss
    cfg, batch, tag = CFG_TINY, 2, "g1_tiny_eval"
    net, sd = _build(cfg)
    net.train()
    x, onehot = _data(cfg, batch, tag)
    crit = DeepSuperloss(criterion=CEPlusDice(weight=None, ignore_index=0))
    opt = FlatAdam(net)
    losses = []
    for _ in range(4):
        opt.zero_grad()
        loss = crit(net(x.to(DEV)), onehot.to(DEV))
        loss.backward()
        opt.step()
        losses.append(loss.item())
    print("  losses", losses)
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    sd2 = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    assert list(sd2.keys()) == list(sd.keys())
    net2, _ = _build(cfg)
    net2.load_state_dict(sd2)
    net.eval(), net2.eval()
    with torch.no_grad():
        a, b = net(x.to(DEV)), net2(x.to(DEV))
    assert _rel(a[0], b[0]) == 0.0            # eval forward is bitwise reproducible (no float atomics)


def test_full_size_forward_vs_reference_golden():
    """BASELINE config #2 shape (4x128^3, n_filters 32, transformer_depth 24), B=1, fp32 path, against the
    strided logits / Dice captured from the real reference (g5_full_eval)."""
    path = os.path.join(GOLDEN, "g5_full_eval.npz")
    if not os.path.exists(path):
        pytest.skip("g5 fixture not generated")
    g = np.load(path)
    cfg = (4, 4, 32, (128, 128, 128), 24)
    net, sd = _build(cfg)
    net.eval()
    x, onehot = _data(cfg, 1, "g5_full_eval")
    with torch.no_grad():
        outs = net(x.to(DEV))
    for i in range(4):
        e = _rel(_strided(outs[i], max(1, int(g["sample_step"]) >> i)), torch.from_numpy(g[f"out{i}"]))
        print(f"  full-size out{i} rel {e:.3e}")
        assert e < 1e-3
    from hdf_rt.loss_fn import compute_dice
    d = compute_dice(outs[0], onehot.to(DEV))
    print("  dice", d, float(g["dice_rounded"]))
    assert abs(d - float(g["dice_rounded"])) <= 1e-4


@pytest.mark.parametrize("tag", ["c4", "c4_absent", "c3"])
def test_running_dice_confusion_matrix_vs_reference_golden(tag):
    """metrics.RunningDice (sklearn confusion matrix on the host in the reference) from on-device counts."""
    from hdf_rt.loss_fn import RunningDice
    g = np.load(os.path.join(GOLDEN, "g7_metric.npz"))
    logits = torch.from_numpy(g[tag + "_logits"]).to(DEV)
    onehot = torch.from_numpy(g[tag + "_onehot"].astype(np.float32)).to(DEV)
    c = logits.shape[1]
    rd = RunningDice(labels=range(c), ignore_label=-1)
    rd.update_from_logits(onehot, logits)
    mean, per = rd.compute_dice()
    assert abs(mean - float(g[tag + "_run_dice"])) < 1e-6
    assert np.allclose(per, g[tag + "_run_list"], atol=1e-4)
    rd.update_from_logits(onehot, logits)                  # running accumulation: doubling every count keeps the ratios
    mean2, _ = rd.compute_dice()
    assert abs(mean2 - mean) < 1e-5
    # the reference's own call signature (metrics.py:104; trainer.py:393-398 passes numpy argmax maps)
    rd2 = RunningDice(labels=range(c), ignore_label=-1)
    rd2.update_matrix(onehot.argmax(1).cpu().numpy(), logits.argmax(1).cpu().numpy())
    mean3, per3 = rd2.compute_dice()
    assert abs(mean3 - float(g[tag + "_run_dice"])) < 1e-6 and np.allclose(per3, g[tag + "_run_list"], atol=1e-4)
    assert hasattr(mean3, "item")
    # metrics.py:122-124: an update whose ground truth is all `ignore_label` is dropped (default ignore_label = 0)
    rd3 = RunningDice(labels=range(c), ignore_label=0)
    rd3.update_from_logits(onehot, logits)
    before = rd3.conf.clone()
    bg = torch.zeros_like(onehot)
    bg[:, 0] = 1.0
    rd3.update_from_logits(bg, logits)
    rd3.update_matrix(np.zeros((2, 8, 8, 8), dtype=np.int64), logits.argmax(1).cpu().numpy())
    assert torch.equal(rd3.conf, before)
    rd3.update_from_logits(onehot, logits)
    assert torch.equal(rd3.conf, 2 * before)
    # a FLOAT class map of shape [B, H, W] is a class map (the reference accepts any dtype), never a score tensor ...
    rd4 = RunningDice(labels=range(c), ignore_label=-1)
    rd4.update_matrix(onehot.argmax(1).float()[:, 0], logits.argmax(1).float()[:, 0])
    rd5 = RunningDice(labels=range(c), ignore_label=-1)
    rd5.update_matrix(onehot.argmax(1)[:, 0].cpu().numpy(), logits.argmax(1)[:, 0].cpu().numpy())
    assert torch.equal(rd4.conf, rd5.conf)
    # ... and ground truth that is all `ignore_label` except for labels >= n_cls is NOT all-ignore (metrics.py:122)
    rd6 = RunningDice(labels=range(c), ignore_label=0)
    gt = np.zeros((2, 8, 8, 8), dtype=np.int64)
    gt[0, 0, 0, 0] = c + 3
    gt[1, 1, 1, 1] = 1
    rd6.update_matrix(gt, logits.argmax(1).cpu().numpy())
    assert int(rd6.conf.sum()) == gt.size - 1 and int(rd6.conf[1].sum()) == 1
    with pytest.raises(Exception):
        rd6.update_from_logits(onehot.argmax(1).float(), logits.argmax(1).float())


@pytest.mark.gpu
def test_two_rank_data_parallel_on_one_gpu():
    """GradSync + staged backward + comm-stream overlap on the real HIP path: two processes share cuda:0 over gloo
    (RCCL refuses two ranks on one device); synced gradient == mean of the local gradients, replicas stay equal."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", "29581", os.path.join(root, "tools", "ddp_check.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    recs = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(recs) == 2, r.stdout
    for rec in recs:
        assert rec["grad_rel_err"] < 1e-5, rec
        assert rec["param_max_diff"] == 0.0, rec


@pytest.mark.gpu
def test_sliding_window_inference_vs_reference_golden():
    """window loop of trainer.py:527-584 on the HIP path (fp32) vs the reference's own result."""
    from hdf_rt.inference import cal_steps, sliding_window_predict
    from models.HDenseFormer import HDenseFormer
    import json
    g = np.load(os.path.join(GOLDEN, "g8_sliding_window.npz"), allow_pickle=False)
    for case in json.loads(str(g["steps_json"])):
        assert cal_steps(tuple(case["size"]), tuple(case["patch"]), tuple(case["step"])) == case["steps"], case
    in_ch, n_cls, nf, td = [int(v) for v in g["cfg"]]
    patch = tuple(int(v) for v in g["patch"])
    step = tuple(int(v) for v in g["step"])
    size = tuple(int(v) for v in g["image_size"])
    net = HDenseFormer(in_ch, n_cls, nf, image_size=patch, transformer_depth=td)
    net.load_state_dict(orc.det_model(in_ch, n_cls, nf, patch, td))
    net = net.cuda()
    net.compute_dtype = "fp32"
    image = detgen.det_input(1, in_ch, size, tag="sw")[0]
    lab, mean = sliding_window_predict(net, image, patch, step, return_probabilities=True)
    mean = mean.cpu().numpy()
    err = np.abs(mean[:, ::2, ::2, ::2] - g["mean_s2"]).max() / np.abs(g["mean_s2"]).max()
    assert err < 1e-3, err                      # mean class probabilities, tolerance of the fp32 logits gate
    agree = (lab.cpu().numpy() == g["argmax"]).mean()
    assert agree > 0.999, agree                 # votes differ only where two mean probabilities tie to ~1e-6
    # windows batched through one forward give the same accumulators as one at a time
    lab1, mean1 = sliding_window_predict(net, image, patch, step, return_probabilities=True, window_batch=1)
    # (batch 1 and batch 4 plans reduce the InstanceNorm partials in different orders: ~3e-6 on the probabilities)
    assert (lab1 == lab).float().mean().item() > 0.9999
    assert float((mean1.cpu() - torch.from_numpy(mean)).abs().max()) < 2e-5
    # a volume shorter than the patch along one axis: clipped window (trainer.py:529-541), zero-padded for the plan
    from oracle import sw_oracle
    small = image[:, :20]
    lab_s, mean_s = sliding_window_predict(net, small, patch, step, return_probabilities=True)
    sd = orc.det_model(in_ch, n_cls, nf, patch, td)

    def fwd(p):
        pad = torch.zeros((1, in_ch) + patch)
        pad[:, :, :p.shape[2], :p.shape[3], :p.shape[4]] = p
        return orc.forward(pad, sd)[0][:, :, :p.shape[2], :p.shape[3], :p.shape[4]]
    with torch.no_grad():
        ref_lab, ref_mean = sw_oracle.sliding_window(fwd, small, n_cls, patch, step)
    assert tuple(lab_s.shape) == (20,) + size[1:]
    assert np.abs(mean_s.cpu().numpy() - ref_mean).max() < 1e-3
    assert (lab_s.cpu().numpy() == ref_lab).mean() > 0.999
    with pytest.raises(ValueError):
        sliding_window_predict(net, image, (16, 32, 32), step)       # patch must be the model's image_size


@pytest.mark.gpu
def test_onehot_from_labels_vs_reference_golden():
    """To_Tensor one-hot (data_loader.py:146-151) expanded on the device from the uint8 map."""
    from hdf_rt.inference import onehot_from_labels
    g = np.load(os.path.join(GOLDEN, "g9_to_tensor.npz"), allow_pickle=False)
    lab = torch.from_numpy(np.stack([g["label"], g["label"][::-1].copy()])).cuda()
    out = onehot_from_labels(lab, int(g["n_cls"])).cpu().numpy()
    assert np.array_equal(out[0], g["onehot"])
    assert np.array_equal(out[1], g["onehot"][:, ::-1])


@pytest.mark.gpu
def test_n_filters_64_forward_backward_vs_oracle():
    """The widest supported model (n_filters = 64: token dim 256, Linear0 up to 352 inputs, 512-channel bottleneck)."""
    from loss.combine_loss import CEPlusDice, DeepSuperloss
    cfg, batch, tag = (2, 3, 64, (32, 32, 32), 8), 1, "nf64"
    net, sd = _build(cfg)
    net.train()
    net.set_dropout_seed(31)
    x, onehot = _data(cfg, batch, tag)
    crit = DeepSuperloss(criterion=CEPlusDice(weight=None, ignore_index=0))
    outs = net(x.to(DEV))
    loss = crit(outs, onehot.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    tr = orc.OracleTrainer(sd)
    ref_loss, ref_outs = tr.loss_and_grads(x, onehot, 31)
    for i in range(4):
        assert _rel(outs[i].detach(), ref_outs[i]) < 1e-3, f"out{i}"
    assert abs(loss.item() - ref_loss.item()) < 1e-4 * max(1.0, abs(ref_loss.item()))
    errs = []
    for name, p in net.named_parameters():
        rg = tr.sd[name].grad
        if rg.norm() < 1e-6:
            continue
        errs.append((name, _rl2(p.grad, rg)))
    errs.sort(key=lambda kv: -kv[1])
    print("  nf64 worst grads", errs[:4])
    assert errs[0][1] < 2e-2, errs[:5]


@pytest.mark.gpu
def test_n_filters_48_forward_backward_vs_oracle():
    """BASELINE config #5 channel widths (n_filters = 48: 96-byte channel rows, 6 / 12 / 24 / 48 chunk lanes per
    voxel, i.e. the non-power-of-two paths of the elementwise and head kernels) on a small volume."""
    from loss.combine_loss import CEPlusDice, DeepSuperloss
    cfg, batch, tag = (2, 3, 48, (32, 32, 32), 4), 1, "nf48"
    net, sd = _build(cfg)
    net.eval()
    x, onehot = _data(cfg, batch, tag)
    crit = DeepSuperloss(criterion=CEPlusDice(weight=None, ignore_index=0))
    outs = net(x.to(DEV))
    loss = crit(outs, onehot.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    tr = orc.OracleTrainer(sd)
    ref_loss, ref_outs = tr.loss_and_grads(x, onehot, None)
    for i in range(4):
        assert _rel(outs[i].detach(), ref_outs[i]) < 1e-3, f"out{i}"
    assert abs(loss.item() - ref_loss.item()) < 1e-4 * max(1.0, abs(ref_loss.item()))
    errs = []
    for name, p in net.named_parameters():
        rg = tr.sd[name].grad
        if rg.norm() < 1e-6:
            continue
        errs.append((name, _rl2(p.grad, rg)))
    errs.sort(key=lambda kv: -kv[1])
    assert errs[0][1] < 1e-2, errs[:5]                       # the reference's own fp32 noise floor (see above)
    d = dict(errs)
    for name in ("conv1x1.weight", "conv1x1_d3.weight", "block_1_2_right.conv.weight"):
        assert d[name] < 5e-3, (name, d[name])


@pytest.mark.gpu
def test_device_normalisation_vs_reference_golden():
    """SURVEY 8f-3, second half: MRNormalize / PETandCTNormalize (data_utils/data_loader.py:39-68) on the device."""
    from hdf_rt.inference import mr_normalize_, pet_ct_normalize_
    g = np.load(os.path.join(GOLDEN, "g10_normalize.npz"), allow_pickle=False)
    mr = mr_normalize_(torch.from_numpy(g["mr_in"]).cuda()).cpu().numpy()
    assert np.array_equal(mr, g["mr_out"])                                   # max and one IEEE division: bit-exact
    pc = pet_ct_normalize_(torch.from_numpy(g["petct_in"]).cuda()).cpu().numpy()
    assert np.array_equal(pc[0], g["petct_out"][0])
    assert np.abs(pc[1] - g["petct_out"][1]).max() < 1e-5                    # mean / std: fp64 here, fp32 pairwise in numpy
    pc2 = pet_ct_normalize_(torch.from_numpy(g["petct_in"]).cuda(), mean=40, w=400).cpu().numpy()
    assert np.abs(pc2 - g["petct_m40_w400_out"]).max() < 1e-5
    with pytest.raises(Exception):
        mr_normalize_(torch.from_numpy(g["mr_in"]))                          # CPU tensor: no fallback


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["c3", "c4_absent"])
def test_standalone_dice_and_ce_losses_vs_plain_torch(tag):
    """loss.dice_loss.DiceLoss(ignore_index=0) and loss.cross_entropy.CrossentropyLoss (reference loss/dice_loss.py:
    53-87, loss/cross_entropy.py:8-22) as stand-alone drop-ins; their sum is CEPlusDice (pinned by g3)."""
    from loss.combine_loss import CEPlusDice
    from loss.cross_entropy import CrossentropyLoss
    from loss.dice_loss import DiceLoss
    g = np.load(os.path.join(GOLDEN, "g3_loss.npz"))
    logits = torch.from_numpy(g[f"{tag}_logits0"])
    onehot = torch.from_numpy(g[tag + "_onehot"].astype(np.float32))
    c = logits.shape[1]
    lr = logits.clone().requires_grad_(True)
    prob = torch.softmax(lr, 1)
    dice = sum((1 - (2 * (prob[:, k] * onehot[:, k]).flatten(1).sum(1) + 1e-5) /
                ((prob[:, k] + onehot[:, k]).flatten(1).sum(1) + 1e-5)).mean() for k in range(1, c)) / (c - 1)
    ce = torch.nn.functional.cross_entropy(lr, onehot.argmax(1))
    gd, = torch.autograd.grad(dice, lr, retain_graph=True)
    gc, = torch.autograd.grad(ce, lr)
    for mod, ref, gref in ((DiceLoss(weight=None, ignore_index=0), dice, gd), (CrossentropyLoss(), ce, gc)):
        x = logits.to(DEV).requires_grad_(True)
        out = mod(x, onehot.to(DEV))
        out.backward()
        assert abs(out.item() - ref.item()) < 2e-5
        assert _rel(x.grad, gref) < 1e-4
    x = logits.to(DEV).requires_grad_(True)
    both = CEPlusDice(weight=None, ignore_index=0)(x, onehot.to(DEV))
    assert abs(both.item() - (dice + ce).item()) < 4e-5


@pytest.mark.gpu
def test_fp16_autocast_with_gradscaler_like_the_reference_trainer():
    """The reference's mixed precision is torch.cuda.amp.autocast(True) = float16 plus a GradScaler
    (trainer.py:20-21,257,369-377).  Same lines here: float16 storage (v_mfma_f32_32x32x16_f16), scaled backward,
    scaler.step() with the optimizer _get_optimizer builds (torch.optim.Adam over the named parameters), and an
    overflowing loss scale must skip the step and back off."""
    from loss.combine_loss import CEPlusDice, DeepSuperloss
    cfg, batch, tag = CFG_TINY, 2, "g1_tiny_eval"
    net, sd = _build(cfg)
    net.eval()                                    # dropout off: compare with the fp32 oracle gradients
    x, onehot = _data(cfg, batch, tag)
    crit = DeepSuperloss(criterion=CEPlusDice(weight=None, ignore_index=0))
    decay, no_decay = orc.param_groups([(k, tuple(v.shape)) for k, v in sd.items()])
    named = dict(net.named_parameters())
    opt = torch.optim.Adam([{"params": [named[k] for k in decay]},
                            {"params": [named[k] for k in no_decay], "weight_decay": 0.0}], lr=1e-3, weight_decay=1e-4)
    scaler = torch.amp.GradScaler("cuda", init_scale=1024.0)
    with torch.autocast(device_type="cuda", dtype=torch.float16):
        outs = net(x.to(DEV))
    assert outs[0].dtype == torch.float16
    loss = crit(outs, onehot.to(DEV))
    opt.zero_grad()
    scaler.scale(loss).backward()
    tr = orc.OracleTrainer(sd)
    ref_loss, ref_outs = tr.loss_and_grads(x, onehot, None)
    assert abs(loss.item() - ref_loss.item()) < 5e-3 * abs(ref_loss.item())
    for i in range(4):
        assert _rl2(outs[i].float(), ref_outs[i]) < 4e-3, f"out{i}"          # 8x tighter than the bf16 gate
    scaler.unscale_(opt)
    mine = torch.cat([p.grad.flatten().cpu() for _, p in net.named_parameters()]).double()
    theirs = torch.cat([tr.sd[n].grad.flatten() for n, _ in net.named_parameters()]).double()
    cos = float((mine @ theirs) / (mine.norm() * theirs.norm()))
    print("  fp16 whole-gradient cosine", cos, "norm ratio", float(mine.norm() / theirs.norm()))
    assert cos > 0.995 and abs(float(mine.norm() / theirs.norm()) - 1) < 2e-2
    before = net.flat_parameters().clone()
    scaler.step(opt)
    scaler.update()
    assert not torch.equal(before, net.flat_parameters())                    # finite gradients: the step was taken
    # overflow: a loss scale beyond the float16 range gives inf gradients -> step skipped, scale halved
    scaler2 = torch.amp.GradScaler("cuda", init_scale=2.0 ** 40)
    with torch.autocast(device_type="cuda", dtype=torch.float16):
        outs = net(x.to(DEV))
    loss = crit(outs, onehot.to(DEV))
    opt.zero_grad()
    scaler2.scale(loss).backward()
    before = net.flat_parameters().clone()
    scaler2.step(opt)
    scaler2.update()
    assert torch.equal(before, net.flat_parameters())
    assert scaler2.get_scale() == 2.0 ** 39


@pytest.mark.gpu
def test_fp16_storage_flat_adam_with_gradscaler():
    """FlatAdam under GradScaler (the scaler walks param_groups[..]['params'] to unscale and to look for infs)."""
    from hdf_rt.optim import FlatAdam
    from loss.combine_loss import CEPlusDice, DeepSuperloss
    cfg, batch, tag = CFG_TINY, 2, "g1_tiny_eval"
    net, sd = _build(cfg, "fp16")
    net.train()
    x, onehot = _data(cfg, batch, tag)
    crit = DeepSuperloss(criterion=CEPlusDice(weight=None, ignore_index=0))
    opt = FlatAdam(net)
    scaler = torch.amp.GradScaler("cuda", init_scale=256.0)
    losses = []
    for _ in range(4):
        opt.zero_grad()
        loss = crit(net(x.to(DEV)), onehot.to(DEV))
        scaler.scale(loss).backward()
        scaler.step(opt)
        scaler.update()
        losses.append(loss.item())
    print("  fp16 losses", losses)
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]


@pytest.mark.gpu
def test_gradsync_bucket_events_keep_the_one_call_backward():
    """under GradSync the backward stays ONE call (hdf_backward_events: branch-stream fork, no host round
    trip between the stages); the library hands back one event per gradient bucket and the communication stream waits
    for them.  Stand-in collective as in the staged test below.  Asserted with HIP events: the five reduces (round 6:
    decoder + heads, encoder levels 1-3, UpConv chain, transformer, encoder level 0 -- hdf_plan_grad_bucket) run in the
    order the buckets become final, the first three START before the backward has finished on the caller's stream (they
    overlap), wait() orders the optimizer behind all of them, and the gradient equals the one-shot backward's."""
    import torch.distributed as dist
    from hdf_rt import parallel
    from loss.combine_loss import CEPlusDice, DeepSuperloss
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29585")
        dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        cfg, batch = (4, 4, 32, (64, 64, 64), 8), 2
        net, sd = _build(cfg, "bf16")
        net.train()
        x, onehot = _data(cfg, batch, "overlap")
        crit = DeepSuperloss(criterion=CEPlusDice(weight=None, ignore_index=0))
        net.set_dropout_seed(5)
        crit(net(x.to(DEV)), onehot.to(DEV)).backward()
        torch.cuda.synchronize()
        g_ref = net.flat_grads().clone()

        marks, sizes = [], []
        spin = int(100e3 * 2.0)

        def fake_allreduce(flat, world, group=None):
            s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s0.record()
            torch.cuda._sleep(spin)
            flat.mul_(1.0)
            s1.record()
            marks.append((s0, s1))
            sizes.append(flat.numel())
        orig = parallel.flat_allreduce_mean
        parallel.flat_allreduce_mean = fake_allreduce
        try:
            sync = parallel.GradSync(net)
            assert not sync.staged
            net.grad_hook = sync
            for p in net.parameters():
                p.grad = None
            net.set_dropout_seed(5)
            t_begin = torch.cuda.Event(enable_timing=True)
            t_begin.record()
            loss = crit(net(x.to(DEV)), onehot.to(DEV))
            loss.backward()
            t_bwd_end = torch.cuda.Event(enable_timing=True)
            t_bwd_end.record()               # caller's stream: everything of the one-call backward is behind this
            sync.wait()
            t_after_wait = torch.cuda.Event(enable_timing=True)
            t_after_wait.record()
            torch.cuda.synchronize()
        finally:
            parallel.flat_allreduce_mean = orig
            net.grad_hook = None
        ms = lambda a, b: a.elapsed_time(b)      # noqa: E731
        assert len(marks) == 5
        ranges = net._last_rt.grad_buckets()
        # the five ranges tile the flat gradient buffer exactly (state_dict order: transformer | chain | encoder level 0 |
        # encoder levels 1-3 | decoder + heads)
        order = sorted(ranges)
        assert order[0][0] == 0 and all(order[i][1] == order[i + 1][0] for i in range(4))
        assert abs(order[-1][1] - net.flat_grads().numel()) < 32
        assert [r for r in order] == [ranges[2], ranges[1], ranges[4], ranges[3], ranges[0]]
        want = [ranges[k][1] - ranges[k][0] for k in sync.EVENT_ORDER]
        assert [abs(a - b) < 32 for a, b in zip(sizes, want)] == [True] * 5, (sizes, want)
        print("  backward end %.3f ms; reduce spans:" % ms(t_begin, t_bwd_end),
              [(round(ms(t_begin, a), 3), round(ms(t_begin, b), 3)) for a, b in marks])
        # the first three buckets (decoder + heads, encoder levels 1-3, UpConv chain) start their reduce while the backward
        # still runs: only the transformer branches' and the 0.1 MB of encoder level 0 are final at its end
        for k in range(3):
            assert ms(marks[k][0], t_bwd_end) > 0.0, k
        for k in range(5):
            assert ms(marks[k][1], t_after_wait) >= 0.0
        assert _rl2(net.flat_grads(), g_ref) < 1e-5
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_gradsync_overlaps_bucket_reduce_with_next_backward_stage():
    """bucket k's all-reduce runs on a side stream while stage k+1 back-propagates.  gloo is host
    synchronous (it cannot overlap) and RCCL needs one GPU per rank, so the collective is replaced by an asynchronous
    device-side stand-in of known length (a spin kernel on the comm stream, like an RCCL kernel would be) and the STREAM
    logic of GradSync is asserted with HIP events: (1) every bucket's reduce starts after its stage and before the
    backward ends, (2) the reduces of buckets 1 and 2 are over before the last stage is (they ran in its shadow),
    (3) wait() orders the optimizer after all of them, (4) the staged gradient equals the one-shot gradient."""
    import torch.distributed as dist
    from hdf_rt import parallel
    from loss.combine_loss import CEPlusDice, DeepSuperloss
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29583")
        dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        cfg, batch = (4, 4, 32, (64, 64, 64), 8), 2
        net, sd = _build(cfg, "bf16")
        net.train()
        x, onehot = _data(cfg, batch, "overlap")
        crit = DeepSuperloss(criterion=CEPlusDice(weight=None, ignore_index=0))
        # reference gradient: one-shot backward, no hook
        net.set_dropout_seed(5)
        crit(net(x.to(DEV)), onehot.to(DEV)).backward()
        torch.cuda.synchronize()
        g_ref = net.flat_grads().clone()

        marks = []
        spin = int(150e3 * 2.0)          # ~150 us at ~2 GHz: comparable with a 26 MB bucket over xGMI

        def fake_allreduce(flat, world, group=None):
            s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s0.record()
            torch.cuda._sleep(spin)
            flat.mul_(1.0)               # touches the bucket on the comm stream, like the collective would
            s1.record()
            marks.append((s0, s1))
        orig = parallel.flat_allreduce_mean
        parallel.flat_allreduce_mean = fake_allreduce
        try:
            sync = parallel.GradSync(net)
            stage_end = []
            calls = []

            def hook(stage):
                e = torch.cuda.Event(enable_timing=True)
                e.record()               # main stream: end of backward stage `stage`
                stage_end.append(e)
                calls.append(stage)
                sync(stage)
            net.grad_hook = hook
            for p in net.parameters():
                p.grad = None
            net.set_dropout_seed(5)
            t_begin = torch.cuda.Event(enable_timing=True)
            t_begin.record()
            crit(net(x.to(DEV)), onehot.to(DEV)).backward()
            sync.wait()
            t_after_wait = torch.cuda.Event(enable_timing=True)
            t_after_wait.record()
            torch.cuda.synchronize()
        finally:
            parallel.flat_allreduce_mean = orig
            net.grad_hook = None
        assert calls == [1, 2, 3] and len(marks) == 3
        ms = lambda a, b: a.elapsed_time(b)      # noqa: E731
        for k in range(3):
            assert ms(stage_end[k], marks[k][0]) >= 0.0                     # reduce k starts after its stage ended
        assert ms(marks[0][0], stage_end[1]) > 0.0 and ms(marks[1][0], stage_end[2]) > 0.0   # ... and while the next runs
        # buckets 1 and 2 finished before the LAST stage did: they were hidden behind backward
        print("  stage ends (ms from begin):", [round(ms(t_begin, e), 3) for e in stage_end],
              "reduce spans:", [(round(ms(t_begin, a), 3), round(ms(t_begin, b), 3)) for a, b in marks])
        assert ms(marks[0][1], stage_end[2]) > 0.0
        assert ms(marks[1][1], stage_end[2]) > -0.05
        for k in range(3):
            assert ms(marks[k][1], t_after_wait) >= 0.0                    # wait(): the main stream continues after them
        # same gradient as the one-shot backward (the transformer / head parameter gradients use fp32 atomics: equal up
        # to summation order)
        assert _rl2(net.flat_grads(), g_ref) < 1e-5
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_bench_two_ranks_on_one_device_exercises_the_multi_gpu_code_path():
    """bench.py --gpus 2 as the driver launches it (torch.distributed.run, one process per rank), with both ranks on
    cuda:0 over gloo (HDF_BENCH_ONE_DEVICE=1; a GPU box here has one device): the barrier + synchronize fences, the MAX
    over ranks of the wall time, GradSync in the step and the single JSON line of rank 0."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HDF_BENCH_ONE_DEVICE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", "29585", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3",
           "--warmup", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                   # exactly one JSON line, from rank 0
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["warmup"] == 1 and rec["scaling"] == "weak"
    assert rec["config"]["global_batch"] == 4 and rec["config"]["parallelism"] == "dp2"
    assert rec["value"] > 0 and abs(rec["value"] - 4 * 3 / (rec["ms_per_step"] * 3e-3)) < 1e-6 * rec["value"]
    assert "roofline" not in rec and "cpu_baseline" not in rec        # N = 1 only


def _torchrun_one_rank(script, args=(), env_extra=None, port=29587, timeout=900):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", **(env_extra or {}))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(root, script), *args]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return [l for l in r.stdout.splitlines() if l.startswith("{")]


@pytest.mark.gpu
def test_rccl_one_rank_preflight_gradsync_over_nccl():
    """The RCCL path itself, as far as one GPU allows (RCCL refuses two ranks on one device): one rank launched by
    torch.distributed.run, init_process_group("nccl", device_id=..), GradSync's broadcast and its three bucket
    all-reduces on the comm stream as RCCL kernels, HSA_ENABLE_IPC_MODE_LEGACY=0.  The synced gradient must equal the
    plain backward's (world 1: sum over one rank / 1) and the optimizer step must run behind sync.wait()."""
    import json
    lines = _torchrun_one_rank(os.path.join("tools", "ddp_check.py"), env_extra={"HDF_DDP_BACKEND": "nccl"}, port=29587)
    assert len(lines) == 1, lines
    rec = json.loads(lines[0])
    assert rec["backend"] == "nccl" and rec["world"] == 1
    assert rec["grad_rel_err"] < 1e-5 and rec["param_max_diff"] == 0.0, rec


@pytest.mark.gpu
def test_bench_under_torchrun_with_one_rank_runs_the_rccl_branch():
    """bench.py exactly as the driver launches the N > 1 runs (torch.distributed.run, backend nccl = RCCL), with
    --nproc-per-node 1: process group on the device, GradSync in the step, barrier + synchronize fences, the MAX
    all-reduce of the wall time on the device, one JSON line."""
    import json
    lines = _torchrun_one_rank("bench.py", args=("--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"),
                               port=29588)
    assert len(lines) == 1, lines
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 1 and rec["config"]["collective"] == "nccl" and rec["config"]["parallelism"] == "dp1"
    assert rec["value"] > 0 and 3.0 < rec["config"]["loss"] < 6.0


@pytest.mark.gpu
def test_staged_backward_with_rccl_matches_the_single_call_backward():
    """tools/stage_cost.py (64^3 here; the 128^3 figures are in profiles/): the three-stage backward + RCCL bucket
    all-reduces gives the gradients of the one-call backward, and reports what the staging costs per step."""
    import json
    lines = _torchrun_one_rank(os.path.join("tools", "stage_cost.py"), port=29589,
                               env_extra={"HDF_STAGE_COST_SIZE": "64", "HDF_STAGE_COST_STEPS": "4"})
    rec = json.loads(lines[-1])
    assert rec["backend"] == "nccl" and rec["grad_rel_err"] < 1e-4, rec
    assert min(rec["ms_three_stages_rccl"]) < 2.0 * min(rec["ms_one_call"]) + 1.0, rec
    # (round 6) the stand-in collective legs ran, and no per-sequence barrier of the persistent kernels gave up beside them
    assert rec["gave_up_workgroup"] == -1, rec
    assert all(k in rec for k in ("ms_one_call_events_standin_rccl_like_32wg_300us", "ms_one_call_events_standin_whole_cu_32wg_300us"))


@pytest.mark.gpu
def test_inference_forward_uses_the_forward_prefix_of_the_workspace():
    """Without an autograd graph (eval / sliding-window prediction) the runtime hands hdf_forward only the prefix of the
    arena a forward touches (hdf_plan_inference_workspace_bytes: no transformer tapes, second dy buffers or 128 MB
    weight-gradient workspace); a later training step re-allocates the full arena, gives the same logits, and a
    backward on the small arena is refused by the library."""
    from hdf_rt._lib import HdfError, check, lib, ptr, stream_ptr
    cfg, batch, tag = (2, 3, 16, (32, 32, 32), 8), 2, "g1_tiny_eval"
    net, _ = _build(cfg)
    net.eval()
    x, onehot = _data(cfg, batch, tag)
    with torch.no_grad():
        outs = [o.float().clone() for o in net(x.to(DEV))]
    rt = net._last_rt
    small = rt.ws.numel()
    assert not rt.ws_backward and small == rt.plan.workspace_bytes(batch, backward=False)
    assert small < 0.6 * rt.plan.workspace_bytes(batch)
    g = torch.zeros(net.flat_parameters().numel(), device=DEV)
    d = [torch.zeros_like(o) for o in outs]
    rc = lib().hdf_backward_stages(rt.plan.h, ptr(x.to(DEV)), ptr(net.flat_parameters()), ptr(rt.ws), rt.ws.numel(),
                                   ptr(d[0]), ptr(d[1]), ptr(d[2]), ptr(d[3]), ptr(g), batch, 7, stream_ptr())
    assert rc != 0 and b"holds a forward only" in lib().hdf_last_error()
    outs2 = net(x.to(DEV))                       # grad mode: the full arena
    assert rt.ws_backward and rt.ws.numel() == rt.plan.workspace_bytes(batch)
    for a, b in zip(outs, outs2):
        assert torch.equal(a, b.detach().float())
    sum(o.float().sum() for o in outs2).backward()
    torch.cuda.synchronize()


@pytest.mark.gpu
def test_replaced_parameters_are_noticed_by_the_next_forward():
    """The forward launches on the flat parameter buffer of the previous call and verifies afterwards that every
    parameter is still a view of it (models/HDenseFormer.py forward): a tensor replaced between two forwards -- a new
    module (`net.conv1x1 = nn.Conv3d(..)`), `p.data = ..`, requires_grad toggled -- must still take effect in the very
    next forward, and the gradients must land on the new tensors."""
    cfg, batch, tag = CFG_TINY, 2, "g1_tiny_eval"
    net, sd = _build(cfg)
    net.eval()
    x, _ = _data(cfg, batch, tag)
    xd = x.to(DEV)
    base = [o.detach().clone() for o in net(xd)]
    again = [o.detach() for o in net(xd)]                     # the optimistic path (second call), nothing changed
    for a, b in zip(base, again):
        assert torch.equal(a, b)
    # 1. a replaced module: the head becomes a fresh Conv3d with other weights
    old_head = net.conv1x1
    new_head = torch.nn.Conv3d(cfg[2], cfg[1], kernel_size=1).to(DEV)
    with torch.no_grad():
        new_head.weight.copy_(old_head.weight * 2.0)
        new_head.bias.copy_(old_head.bias + 1.0)
    net.conv1x1 = new_head
    out = net(xd)
    ref = base[0].float() * 2.0 - old_head.bias.view(1, -1, 1, 1, 1) * 2.0 + (old_head.bias + 1.0).view(1, -1, 1, 1, 1)
    assert _rel(out[0].detach().float(), ref) < 1e-5
    for a, b in zip(base[1:], out[1:]):
        assert torch.equal(a, b.detach())
    # the gradient reaches the NEW tensors
    out[0].float().sum().backward()
    assert net.conv1x1.weight.grad is not None and float(net.conv1x1.weight.grad.abs().sum()) > 0
    assert net.conv1x1.weight.data_ptr() != new_head.weight.data_ptr() or net.conv1x1.weight is new_head.weight
    # 2. p.data swapped behind the module's back
    net.zero_grad(set_to_none=True)
    net(xd)
    net.conv1x1_d1.bias.data = net.conv1x1_d1.bias.data.clone() + 3.0
    out2 = net(xd)
    assert _rel(out2[1].detach().float(), base[1].float() + 3.0) < 1e-5
    # 3. requires_grad switched off everywhere: the next forward must not build a graph
    for p in net.parameters():
        p.requires_grad_(False)
    out3 = net(xd)
    assert not out3[0].requires_grad
    for p in net.parameters():
        p.requires_grad_(True)
    assert net(xd)[0].requires_grad


@pytest.mark.gpu
def test_six_training_steps_follow_the_oracle_trajectory():
    """Six consecutive optimisation steps in train mode (hash dropout on, FlatAdam, zero_grad, the forward that
    launches on the previous step's flat buffer): the loss of every step and the parameters after the last one stay
    with the oracle's (float64) trajectory.  A single-step test cannot see a stale parameter buffer, a dropout seed that
    does not advance, or optimizer state that is lost between steps."""
    from hdf_rt.optim import FlatAdam
    from loss.combine_loss import CEPlusDice, DeepSuperloss
    cfg, batch, tag = CFG_TINY, 2, "g1_tiny_eval"
    net, sd = _build(cfg)
    net.train()
    x, onehot = _data(cfg, batch, tag)
    crit = DeepSuperloss(criterion=CEPlusDice(weight=None, ignore_index=0))
    opt = FlatAdam(net, lr=1e-3, weight_decay=1e-4)
    tr = orc.OracleTrainer({k: v.double() for k, v in sd.items()}, lr=1e-3, weight_decay=1e-4)
    xd, od = x.to(DEV), onehot.to(DEV)
    for step in range(6):
        seed = net.step_seed(net._step + 1, 0)
        opt.zero_grad()
        loss = crit(net(xd), od)
        loss.backward()
        opt.step()
        ref_loss, _ = tr.step(x.double(), onehot.double(), seed)
        print(f"  step {step}: loss {loss.item():.6f}  oracle {ref_loss.item():.6f}")
        assert abs(loss.item() - ref_loss.item()) < 2e-3 * abs(ref_loss.item()), step
    errs = sorted(((_rl2(p.detach(), tr.sd[name].detach()), name) for name, p in net.named_parameters()), reverse=True)
    print("  worst parameters after 6 steps:", errs[:3])
    # Adam's first steps move every entry by ~lr whatever the gradient's size, so an entry whose gradient sign is decided
    # by rounding ends 2 lr per step away from the oracle's (weights are ~3e-2 in size): measured worst tensor 1.1e-2 in
    # norm (deep_conv, whose gradient sits behind every ReLU / max-pool decision); a stale buffer or a lost optimizer
    # state is an O(1) difference
    assert errs[0][0] < 3e-2
