"""The persistent transformer kernels (csrc/transformer_chain.hip: every dense layer of every block in one launch per
direction, per-sequence barriers) against the launch chain they replace (tok_fwd / attention / tok_bwd ..., selected with
HDF_NO_TF_CHAIN=1): the same fp32 operations in the same order, so every saved tensor of the branches
(HDenseFormer.py:78-145: h0, q|k|v, attention output, log-sum-exp, h1, h2, the feature buffers), the four outputs and --
in backward -- every gradient must agree BIT FOR BIT, and no per-sequence barrier may have timed out."""
import os

import pytest
import torch

from hdf_rt._lib import BF16, F16, F32
from hdf_rt.runtime import Plan, Runtime

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

# (in_channels, n_cls, n_filters, image, depth, batch, dtype)
CASES = [
    (2, 3, 16, (32, 32, 32), 8, 2, F32),        # N = 8: one partial tile per sequence
    (1, 2, 16, (48, 48, 48), 4, 3, F32),        # N = 27: two tiles, the second partial; odd batch
    (4, 4, 32, (64, 64, 64), 8, 2, BF16),       # N = 64
    (4, 4, 32, (128, 128, 128), 24, 2, BF16),   # BASELINE configs[1]: 256 workgroups
    (4, 4, 32, (128, 128, 128), 24, 1, F32),
    (2, 3, 32, (144, 144, 144), 8, 1, BF16),    # configs[3] geometry: N = 729 (46 tiles, the last partial)
    (4, 4, 48, (160, 160, 160), 4, 1, F16),     # configs[4] geometry: N = 1000, token dim 192
]


def _params(plan, seed):
    g = torch.Generator().manual_seed(seed)
    flat = torch.zeros(plan.param_floats)
    for name, off, numel, shape in plan.table:
        if name.endswith("norm.weight") or (name.endswith(".weight") and len(shape) == 1):
            v = 1.0 + 0.1 * torch.randn(numel, generator=g)
        elif len(shape) > 1:
            fan = 1
            for s in shape[1:]:
                fan *= s
            v = torch.randn(numel, generator=g) / fan ** 0.5
        else:
            v = 0.1 * torch.randn(numel, generator=g)
        flat[off:off + numel] = v
    return flat.to(DEV)


def _forward(case, chain, training=1):
    cin, ncls, nf, image, depth, batch, dtype = case
    if chain:
        os.environ.pop("HDF_NO_TF_CHAIN", None)
    else:
        os.environ["HDF_NO_TF_CHAIN"] = "1"
    try:
        plan = Plan(cin, ncls, nf, image, depth, dtype)
        rt = Runtime(plan, DEV)
        params = _params(plan, 1)
        x = torch.rand((batch, cin) + image, generator=torch.Generator().manual_seed(2)).to(DEV)
        outs = rt.forward(x, params, training, 1234, need_backward=True)
        torch.cuda.synchronize()
        res = {"outs": [o.clone() for o in outs], "F": rt.read_region("tf_F").clone(),
               "save": rt.read_region("tf_save").clone(), "attnall": rt.read_buffer("attnall"),
               "sync": rt.read_region("tf_sync").view(torch.int32).clone()}
        return res, (plan, rt, params, x, outs)
    finally:
        os.environ.pop("HDF_NO_TF_CHAIN", None)


@pytest.mark.parametrize("case", CASES, ids=lambda c: "in%d_nf%d_%d_td%d_b%d_t%d" % (c[0], c[2], c[3][0], c[4], c[5], c[6]))
def test_chain_forward_is_bit_identical_to_the_launch_chain(case):
    ref, _ = _forward(case, chain=False)
    got, _ = _forward(case, chain=True)
    nseq = case[0] * case[5]
    assert int(got["sync"][32 * nseq]) == 0, "a per-sequence barrier of the forward chain timed out"
    ntile = (case[3][0] // 16) ** 3
    ntile = (ntile + 15) // 16
    depth_layers = (case[4] // 4) * 4
    assert [int(got["sync"][32 * s]) for s in range(nseq)] == [ntile * depth_layers] * nseq
    assert int(ref["sync"][0]) == 0 or True   # (the launch chain leaves the counters alone)
    for name in ("save", "F"):
        a, b = got[name], ref[name]
        assert torch.equal(a.view(torch.int32), b.view(torch.int32)), \
            "%s differs: %d of %d words, max |d| %.3e" % (name, int((a.view(torch.int32) != b.view(torch.int32)).sum()),
                                                           a.numel(), float((a - b).abs().max()))
    assert torch.equal(got["attnall"], ref["attnall"])
    for a, b in zip(got["outs"], ref["outs"]):
        assert torch.equal(a.view(torch.int16 if a.element_size() == 2 else torch.int32),
                           b.view(torch.int16 if b.element_size() == 2 else torch.int32))


def test_chain_forward_eval_mode_and_repeat_calls():
    """dropout off; and a second call on the same workspace (the counters are re-zeroed by every launch)"""
    case = CASES[2]
    ref, _ = _forward(case, chain=False, training=0)
    got, (plan, rt, params, x, _) = _forward(case, chain=True, training=0)
    assert torch.equal(got["save"].view(torch.int32), ref["save"].view(torch.int32))
    outs2 = rt.forward(x, params, 0, 1234, need_backward=True)
    torch.cuda.synchronize()
    for a, b in zip(outs2, got["outs"]):
        assert torch.equal(a, b)
    assert int(rt.read_region("tf_sync").view(torch.int32)[32 * case[0] * case[5]]) == 0


def _backward(case, chain):
    """forward + backward under one arrangement; returns the gradient buffer and the transformer's backward regions"""
    cin, ncls, nf, image, depth, batch, dtype = case
    res, (plan, rt, params, x, outs) = _forward(case, chain)
    if chain:
        os.environ.pop("HDF_NO_TF_CHAIN", None)
    else:
        os.environ["HDF_NO_TF_CHAIN"] = "1"
    try:
        g = torch.Generator().manual_seed(5)
        douts = [(torch.randn(o.shape, generator=g) * 1e-2).to(DEV).to(o.dtype) for o in outs]
        grads = torch.zeros_like(params)
        rt.backward(x, params, douts, grads)
        torch.cuda.synchronize()
        N = (image[0] // 16) ** 3
        rows = cin * batch * N
        DMF = 4 * nf + 128
        return {"grads": grads.clone(), "tape": rt.read_region("tf_tape").clone(),
                "otape": rt.read_region("tf_otape").clone(),
                "dF": rt.read_region("tf_dF").view(rows, DMF)[:, :4 * nf].clone(),
                "sync": rt.read_region("tf_sync").view(torch.int32).clone(), "table": plan.table}
    finally:
        os.environ.pop("HDF_NO_TF_CHAIN", None)


@pytest.mark.parametrize("case", CASES, ids=lambda c: "in%d_nf%d_%d_td%d_b%d_t%d" % (c[0], c[2], c[3][0], c[4], c[5], c[6]))
def test_chain_backward_is_bit_identical_to_the_launch_chain(case):
    """tapes (the operands of every weight-matrix gradient), the input gradient of block 0 and every gradient that is a
    fixed-order sum: bit for bit; bias / LayerNorm-parameter gradients (fp32 atomics in both arrangements): 1e-4"""
    ref = _backward(case, chain=False)
    got = _backward(case, chain=True)
    nseq = case[0] * case[5]
    half = 1 << 17
    assert int(got["sync"][32 * nseq]) == 0 and int(got["sync"][half + 32 * nseq]) == 0, "a per-sequence barrier timed out"
    for name in ("dF", "tape", "otape"):
        a, b = got[name], ref[name]
        nd = a.view(torch.int32) != b.view(torch.int32)
        assert not bool(nd.any()), "%s differs in %d of %d words, max |d| %.3e" % (
            name, int(nd.sum()), a.numel(), float((a - b).abs().max()))
    for name, off, numel, shape in got["table"]:
        a, b = got["grads"][off:off + numel], ref["grads"][off:off + numel]
        # bit for bit: the weight matrices of the branches (fixed-order sums over the tapes).  Everything else to 1e-4: the
        # branches' bias / LayerNorm / embedding gradients and the heads' are fp32 atomics in both arrangements, and the
        # U-Net's gradients do not depend on the arrangement at all (some of its small-shape reductions are not
        # reproducible run to run either, which is why they are not compared exactly)
        exact = name.startswith("attns.") and len(shape) == 2 and "patch" not in name
        if exact:
            assert torch.equal(a.view(torch.int32), b.view(torch.int32)), name
        else:
            scale = float(b.abs().max()) + 1e-12
            assert float((a - b).abs().max()) <= 1e-4 * scale + 1e-9, name


def test_a_persistent_launch_that_cannot_be_resident_gives_up_cleanly_and_the_plan_falls_back():
    """a per-sequence barrier that is not completed in time no longer traps (a trap kills the caller's HIP
    context).  The grid of this geometry -- 8 modalities x 2 samples x 32 tiles = 512 workgroups -- cannot be resident on
    256 compute units; hdf_plan_force_persistent (tests only) makes the plan launch it anyway with a 3 ms deadline.  The
    resident half waits for siblings that cannot be dispatched and gives up: the launch ends by itself, the device stays
    usable, the outputs of that call are NaN (never plausible garbage), the plan's NEXT call reports
    HDF_ERR_CHAIN_TIMEOUT once without launching anything, and from then on the plan runs the launch chain -- the same
    results as a plan that never tried."""
    import ctypes as C
    from hdf_rt._lib import HdfError, check, lib
    case = (8, 2, 16, (128, 128, 128), 4, 2, BF16)
    cin, ncls, nf, image, depth, batch, dtype = case
    assert cin * batch * 32 > torch.cuda.get_device_properties(0).multi_processor_count
    good, _ = _forward(case, chain=True)        # (not forced: 512 tiles > 256 units, i.e. the launch chain)
    plan = Plan(cin, ncls, nf, image, depth, dtype)
    rt = Runtime(plan, DEV)
    params = _params(plan, 1)
    x = torch.rand((batch, cin) + image, generator=torch.Generator().manual_seed(2)).to(DEV)
    pers, who = C.c_int(-1), C.c_int(-2)
    check(lib().hdf_plan_chain_state(plan.h, batch, C.byref(pers), C.byref(who)), "chain_state")
    assert (pers.value, who.value) == (0, -1)
    check(lib().hdf_plan_force_persistent(plan.h, 1), "force_persistent")
    check(lib().hdf_plan_set_chain_timeout_us(plan.h, 3000), "set_chain_timeout")
    check(lib().hdf_plan_chain_state(plan.h, batch, C.byref(pers), C.byref(who)), "chain_state")
    assert pers.value == 1
    outs = rt.forward(x, params, 1, 1234, need_backward=True)
    torch.cuda.synchronize()                      # returns: no trap, no HIP error
    sync = rt.read_region("tf_sync").view(torch.int32)
    assert int(sync[32 * cin * batch]) != 0, "a 512-workgroup persistent grid was expected to give up on 256 units"
    check(lib().hdf_plan_chain_state(plan.h, batch, C.byref(pers), C.byref(who)), "chain_state")
    assert pers.value == 0 and 0 <= who.value < 512, (pers.value, who.value)
    assert bool(torch.isnan(outs[0].float()).any()), "a launch that gave up must poison its outputs"
    with pytest.raises(HdfError, match="gave up"):
        rt.forward(x, params, 1, 1234, need_backward=True)
    outs3 = rt.forward(x, params, 1, 1234, need_backward=True)    # the launch chain from now on, forced or not
    torch.cuda.synchronize()
    check(lib().hdf_plan_chain_state(plan.h, batch, C.byref(pers), C.byref(who)), "chain_state")
    assert pers.value == 0
    for a, b in zip(outs3, good["outs"]):
        assert torch.equal(a, b)
    assert torch.equal(rt.read_region("tf_save").view(torch.int32), good["save"].view(torch.int32))
    # ... and a backward behind that forward follows it (launch chain): finite gradients
    g = torch.Generator().manual_seed(5)
    douts = [(torch.randn(o.shape, generator=g) * 1e-2).to(DEV).to(o.dtype) for o in outs3]
    grads = torch.zeros_like(params)
    rt.backward(x, params, douts, grads)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(grads).all())


def test_the_backward_follows_the_arrangement_its_forward_ran():
    """HDF_NO_TF_CHAIN is read once per FORWARD; flipping it between the forward and its backward must not
    make the persistent backward consume operand records the launch-chain forward never wrote."""
    case = CASES[2]
    ref = _backward(case, chain=False)
    res, (plan, rt, params, x, outs) = _forward(case, chain=False)     # launch-chain forward ...
    os.environ.pop("HDF_NO_TF_CHAIN", None)                             # ... and the knob is gone before the backward
    g = torch.Generator().manual_seed(5)
    douts = [(torch.randn(o.shape, generator=g) * 1e-2).to(DEV).to(o.dtype) for o in outs]
    grads = torch.zeros_like(params)
    rt.backward(x, params, douts, grads)
    torch.cuda.synchronize()
    assert torch.equal(rt.read_region("tf_tape").view(torch.int32), ref["tape"].view(torch.int32))
    for name, off, numel, shape in plan.table:
        if name.startswith("attns.") and len(shape) == 2 and "patch" not in name:
            assert torch.equal(grads[off:off + numel].view(torch.int32), ref["grads"][off:off + numel].view(torch.int32)), name


def test_two_plans_on_two_streams_take_turns_with_their_persistent_launches():
    """two persistent launches resident together could each hold a part of the compute units and wait for
    siblings that are never dispatched.  The library orders every persistent launch of the process behind the previous one
    (an event chain per device), so two plans at the 256-workgroup geometry driven from two streams both finish, without a
    give-up, with the results of a solo run."""
    import ctypes as C
    from hdf_rt._lib import check, lib
    case = (4, 4, 32, (128, 128, 128), 8, 2, BF16)
    cin, ncls, nf, image, depth, batch, dtype = case
    solo, (plan_a, rt_a, params, x, _) = _forward(case, chain=True, training=0)
    plan_b = Plan(cin, ncls, nf, image, depth, dtype)
    rt_b = Runtime(plan_b, DEV)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    res = []
    for _ in range(3):
        with torch.cuda.stream(sa):
            oa = rt_a.forward(x, params, 0, 1234, need_backward=False)
        with torch.cuda.stream(sb):
            ob = rt_b.forward(x, params, 0, 1234, need_backward=False)
        res.append((oa, ob))
    torch.cuda.synchronize()
    pers, who = C.c_int(), C.c_int()
    for pl in (plan_a, plan_b):
        check(lib().hdf_plan_chain_state(pl.h, batch, C.byref(pers), C.byref(who)), "chain_state")
        assert (pers.value, who.value) == (1, -1)
    for oa, ob in res:
        for a, b, c in zip(oa, ob, solo["outs"]):
            assert torch.equal(a, c) and torch.equal(b, c)
