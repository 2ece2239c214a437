"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol declared in
include/hdf.h, the plan's parameter table equals the reference state_dict contract, the drop-in module
keeps the reference's surface, and the product path fails loudly without a GPU (no fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT
from oracle import hdf_oracle as orc


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "hdf.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(hdf_[a-z0-9_]+)\s*\(", hdr)))


@pytest.fixture(scope="module")
def lib():
    from hdf_rt import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import importlib.util
        spec = importlib.util.spec_from_file_location("hdf_build", os.path.join(ROOT, "h-denseformer_amd", "build.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.build(verbose=False)
    return _lib.lib()


def test_library_exports_every_declared_symbol(lib):
    from hdf_rt import _lib
    declared = _declared_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/hdf.h but not exported"
    assert sorted(_lib.EXPORTS) == declared, "ctypes prototypes out of sync with include/hdf.h"
    assert b"gfx950" in lib.hdf_version()


@pytest.mark.parametrize("cfg", [(4, 4, 32, (128, 128, 128), 24), (2, 3, 16, (32, 32, 32), 8),
                                 (2, 3, 32, (144, 144, 144), 24), (4, 4, 48, (160, 160, 160), 24)])
def test_plan_param_table_matches_reference_state_dict(lib, cfg):
    from hdf_rt import _lib
    from hdf_rt.runtime import Plan
    plan = Plan(*cfg, _lib.F32)
    shapes = orc.state_dict_shapes(*cfg)
    assert [t[0] for t in plan.table] == list(shapes.keys())
    for (name, off, numel, shape), ref in zip(plan.table, shapes.values()):
        assert tuple(shape) == tuple(ref), name
        assert off % 16 == 0 and numel == int(np.prod(ref))
    # modality blocks are equally spaced (the kernels address branch m as base + m*stride)
    o0 = dict((t[0], t[1]) for t in plan.table)
    for m in range(1, cfg[0]):
        assert (o0[f"attns.{m}.position_embeddings"] - o0[f"attns.{m-1}.position_embeddings"]
                == o0["attns.1.position_embeddings"] - o0["attns.0.position_embeddings"])
    assert plan.workspace_bytes(2) > 0


def test_plan_rejects_bad_configs(lib):
    from hdf_rt import _lib
    h = C.c_void_p()
    for bad in [(4, 4, 32, 120, 128, 128, 24), (4, 4, 20, 128, 128, 128, 24), (4, 4, 32, 16, 16, 16, 24),
                (4, 9, 32, 128, 128, 128, 24)]:
        rc = lib.hdf_plan_create(*bad, _lib.F32, C.byref(h))
        assert rc != 0 and len(lib.hdf_last_error()) > 0


def test_dropin_module_surface():
    from models.HDenseFormer import HDenseFormer, HDenseFormer_16, HDenseFormer_32
    net = HDenseFormer_16(in_channels=2, n_cls=3, image_size=(32, 32, 32), transformer_depth=8)
    shapes = orc.state_dict_shapes(2, 3, 16, (32, 32, 32), 8)
    sd = net.state_dict()
    assert list(sd.keys()) == list(shapes.keys())
    assert all(tuple(sd[k].shape) == tuple(v) for k, v in shapes.items())
    assert len(list(net.buffers())) == 0
    assert [n for n, _ in net.named_parameters()] == list(shapes.keys())
    # optimizer grouping rule of trainer.py:812-817 depends only on names/shapes
    decay, no_decay = orc.param_groups([(n, tuple(p.shape)) for n, p in net.named_parameters()])
    mask = net.weight_decay_mask()
    tbl = {t[0]: t for t in net._plan(0).table}
    for n in decay[:5] + no_decay[:5]:
        _, off, numel, _s = tbl[n]
        assert int(mask[off: off + numel].sum()) == (numel if n in decay else 0)
    # torch default init statistics are those of the reference's layers (zeros pos-emb, unit norms)
    assert float(sd["attns.0.position_embeddings"].abs().max()) == 0.0
    assert float((sd["block_1_1_left.norm.weight"] - 1).abs().max()) == 0.0
    # flat buffer aliasing survives load_state_dict
    det = orc.det_model(2, 3, 16, (32, 32, 32), 8)
    flat = net.flat_parameters()
    net.load_state_dict(det)
    assert net.flat_parameters() is flat
    _, off, numel, shape = tbl["conv1x1_d2.weight"]
    assert torch.equal(flat[off: off + numel].view(shape), det["conv1x1_d2.weight"])
    assert isinstance(HDenseFormer_32(4, 4, (32, 32, 32), 4), HDenseFormer)


def test_product_path_fails_loudly_without_gpu():
    from hdf_rt import _lib
    from loss.combine_loss import CEPlusDice, DeepSuperloss
    from models.HDenseFormer import HDenseFormer_16
    net = HDenseFormer_16(in_channels=2, n_cls=3, image_size=(32, 32, 32), transformer_depth=8)
    with pytest.raises(_lib.HdfError):
        net(torch.zeros(1, 2, 32, 32, 32))
    crit = DeepSuperloss(criterion=CEPlusDice(weight=None, ignore_index=0))
    with pytest.raises(_lib.HdfError):
        crit([torch.zeros(1, 3, 8, 8, 8)], torch.zeros(1, 3, 8, 8, 8))
    with pytest.raises(NotImplementedError):     # BinaryDiceLoss settings the kernels do not implement
        DeepSuperloss(criterion=CEPlusDice(weight=None, ignore_index=0, p=2))([torch.zeros(1)], torch.zeros(1))


def test_multi_device_dataparallel_is_refused_with_a_pointer_to_the_process_per_gpu_path():
    """trainer.py:228-229 wraps the net in nn.DataParallel when DEVICE lists several GPUs: replication (what
    torch.nn.parallel.replicate calls on every module) must fail loudly and say what to do instead; a wrapper around a
    single device never replicates."""
    from hdf_rt import _lib
    from models.HDenseFormer import HDenseFormer_16
    net = HDenseFormer_16(in_channels=2, n_cls=3, image_size=(32, 32, 32), transformer_depth=8)
    with pytest.raises(_lib.HdfError, match="one process per GPU"):
        net._replicate_for_data_parallel()
    wrapped = torch.nn.DataParallel(net)            # no GPU here: device_ids == [] -> plain module call
    assert wrapped.module is net
    with pytest.raises(_lib.HdfError, match="GPU tensor"):
        wrapped(torch.zeros(1, 2, 32, 32, 32))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "h-denseformer_amd")
    for dp, _dn, fn in os.walk(pkg):
        for f in fn:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, os.path.join(dp, f)


@pytest.mark.parametrize("cfg", [(4, 4, 32, (128, 128, 128), 24), (2, 3, 16, (32, 32, 32), 8), (4, 4, 48, (160, 160, 160), 24)])
def test_gradient_buckets_tile_the_flat_buffer_in_state_dict_order(lib, cfg):
    """hdf_plan_grad_bucket (round 6: five buckets of hdf_backward_events): contiguous ranges that tile the flat gradient
    buffer exactly, in state_dict order transformer | UpConv chain | encoder level 0 | encoder levels 1-3 | decoder +
    heads, every tensor whole inside one bucket (host-side only: no GPU needed)."""
    from hdf_rt import _lib
    from hdf_rt.runtime import Plan
    plan = Plan(*cfg, _lib.F32)
    rng = []
    for k in range(5):
        lo, hi = C.c_int64(), C.c_int64()
        assert lib.hdf_plan_grad_bucket(plan.h, k, C.byref(lo), C.byref(hi)) == 0
        rng.append((lo.value, hi.value))
    order = [rng[2], rng[1], rng[4], rng[3], rng[0]]
    assert order[0][0] == 0 and order[-1][1] == plan.param_floats
    assert all(order[i][1] == order[i + 1][0] for i in range(4))
    want = {2: "attns.", 1: ("deep_conv.", "up1.", "up2.", "up3."), 4: ("block_1_1_left.", "block_1_2_left."),
            3: ("block_2_", "block_3_", "block_4_"), 0: ("upconv_", "block_", "conv1x1")}
    for name, off, numel, _shape in plan.table:
        k = next(i for i, (lo, hi) in enumerate(rng) if lo <= off < hi)
        assert off + numel <= rng[k][1], name
        assert name.startswith(want[k]), (name, k)
        if k == 3:
            assert "_left." in name
        if k == 0 and name.startswith("block_"):
            assert "_right." in name
    lo, hi = C.c_int64(), C.c_int64()
    assert lib.hdf_plan_grad_bucket(plan.h, 5, C.byref(lo), C.byref(hi)) != 0
    # 2-D plans report one bucket (their gradients are extracted in one pass): the ranges are refused
    plan2 = Plan(2, 2, 16, (64, 64), 8, _lib.F32)
    assert lib.hdf_plan_grad_bucket(plan2.h, 0, C.byref(lo), C.byref(hi)) != 0


def test_persistent_kernel_controls_on_the_host(lib):
    """the round-6 controls of the persistent transformer kernels are host-side state of the plan: deadline bounds, the
    residency rule (tiles <= compute units: BASELINE configs[1] is exactly 256, configs[4] at batch 2 is 504), the forcing
    hook of the tests, no pending give-up on a fresh plan"""
    from hdf_rt import _lib
    from hdf_rt.runtime import Plan
    plan = Plan(4, 4, 32, (128, 128, 128), 24, _lib.BF16)
    pers, who = C.c_int(-1), C.c_int(-2)
    assert lib.hdf_plan_chain_state(plan.h, 2, C.byref(pers), C.byref(who)) == 0
    assert (pers.value, who.value) == (1, -1)
    assert lib.hdf_plan_chain_state(plan.h, 3, C.byref(pers), C.byref(who)) == 0 and pers.value == 0      # 384 tiles
    assert lib.hdf_plan_set_chain_timeout_us(plan.h, 50) != 0 and lib.hdf_plan_set_chain_timeout_us(plan.h, 5000) == 0
    big = Plan(4, 4, 48, (160, 160, 160), 24, _lib.F16)
    assert lib.hdf_plan_chain_state(big.h, 1, C.byref(pers), C.byref(who)) == 0 and pers.value == 1       # 252 tiles
    assert lib.hdf_plan_chain_state(big.h, 2, C.byref(pers), C.byref(who)) == 0 and pers.value == 0       # 504 tiles
    assert lib.hdf_plan_force_persistent(big.h, 1) == 0
    assert lib.hdf_plan_chain_state(big.h, 2, C.byref(pers), C.byref(who)) == 0 and pers.value == 1
    assert lib.hdf_set_cu_budget(128) == 0
    try:
        assert lib.hdf_plan_chain_state(plan.h, 2, C.byref(pers), C.byref(who)) == 0 and pers.value == 0   # 256 tiles > 128
    finally:
        assert lib.hdf_set_cu_budget(256) == 0
