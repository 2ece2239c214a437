"""Drop-in `models.HDenseFormer` for MI355X: same constructor, factories, state_dict keys and 4-output
forward as the reference (models/HDenseFormer.py:177-261), with all arithmetic in libhdf_hip.so.

The nn.Module tree below exists ONLY to own parameters under the reference's names (SURVEY.md appendix
C: e.g. `attns.0.blocks.2.0.layers.1.1.fn.to_qkv.weight`) and to reproduce torch's default
initialisation; none of the holder modules is ever called.  `forward` hands x and ONE flat fp32
parameter buffer to the C ABI (hdf_forward / hdf_backward in include/hdf.h) through a single autograd
node.  Precision: fp32 storage + exact-fp32 MFMA by default (like the reference without AMP); under
`torch.autocast(device_type="cuda", dtype=torch.bfloat16)` (trainer.py:369 `autocast`) or with
`net.compute_dtype = "bf16"` activations are stored in bf16 and the matrix cores run bf16 with fp32
accumulation, and the four outputs come back in bf16 exactly as autocast would return them.  The
reference's own mixed precision is `torch.cuda.amp.autocast(True)` = float16 with a GradScaler
(trainer.py:20-21,257,369-377): under float16 autocast (or `compute_dtype = "fp16"`) storage is IEEE
half with v_mfma_f32_32x32x16_f16 (same rate as bf16), outputs come back in float16, and the scaled
backward runs unchanged (parameter gradients stay fp32, so GradScaler.unscale_/step see what they expect).

There is no CPU or eager fallback: without the built extension or on a non-GPU tensor, forward raises.
"""
import torch
from torch import nn

from hdf_rt import _lib
from hdf_rt.runtime import HDFFunction, Plan, Runtime

__all__ = ["HDenseFormer", "HDenseFormer_32", "HDenseFormer_16"]

_GROWTH, _HEADS, _LAYERS, _PATCH = 32, 8, 4, 16


# ------------------------------------------------------------------ parameter holders (never called)
class _Holder(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter holder: the computation lives in libhdf_hip.so (see HDenseFormer.forward)")


class _Normed(_Holder):          # PreNorm: .norm + .fn
    def __init__(self, width, fn):
        super().__init__()
        self.norm = nn.LayerNorm(width)
        self.fn = fn


class _TwoLinear(_Holder):       # DenseForward: net.0 / net.3 are the Linears
    def __init__(self, d_in, d_hidden, d_out):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(d_in, d_hidden), nn.GELU(), nn.Dropout(0.5), nn.Linear(d_hidden, d_out),
                                 nn.Dropout(0.5))


class _QKVOut(_Holder):          # Dense_Attention: to_qkv (no bias), to_out.0
    def __init__(self, width):
        super().__init__()
        self.to_qkv = nn.Linear(width, 3 * width, bias=False)
        self.to_out = nn.Sequential(nn.Linear(width, width), nn.Dropout(0.5))


class _DenseBlock(_Holder):
    def __init__(self, dim):
        super().__init__()
        self.layers = nn.ModuleList([
            nn.ModuleList([nn.Linear(dim + i * _GROWTH, _GROWTH), _Normed(_GROWTH, _QKVOut(_GROWTH)),
                           _Normed(_GROWTH, _TwoLinear(_GROWTH, 2 * _GROWTH, _GROWTH))]) for i in range(_LAYERS)])
        self.out_layer = _TwoLinear(dim + _LAYERS * _GROWTH, 2 * _GROWTH, dim)


_CONV = {2: nn.Conv2d, 3: nn.Conv3d}
_CONVT = {2: nn.ConvTranspose2d, 3: nn.ConvTranspose3d}
_INORM = {2: nn.InstanceNorm2d, 3: nn.InstanceNorm3d}


class _Branch(_Holder):
    def __init__(self, dim, tokens, n_blocks, nd=3):
        super().__init__()
        self.patch_embeddings = _CONV[nd](1, dim, kernel_size=_PATCH, stride=_PATCH)
        self.position_embeddings = nn.Parameter(torch.zeros(1, tokens, dim))
        self.blocks = nn.ModuleList([nn.ModuleList([_DenseBlock(dim)]) for _ in range(n_blocks)])


class _ConvNormAct(_Holder):     # BasicConv3d / BasicConv2d: conv (no bias) + affine InstanceNorm
    def __init__(self, cin, cout, nd=3):
        super().__init__()
        self.conv = _CONV[nd](cin, cout, kernel_size=3, stride=1, padding=1, bias=False)
        self.norm = _INORM[nd](cout, affine=True)


class _ConvUp(_Holder):          # UpConv: double_conv.0 is the conv (with bias); the norm has no parameters
    def __init__(self, cin, cout, nd=3):
        super().__init__()
        self.double_conv = nn.Sequential(_CONV[nd](cin, cout, kernel_size=3, padding=1), _INORM[nd](cout),
                                         nn.ReLU(inplace=True))


# ------------------------------------------------------------------------------------------- model
class HDenseFormer(nn.Module):
    _ND = 3          # models/HDenseFormer_2D.py derives the 2-D model from this class with _ND = 2

    def __init__(self, in_channels, n_cls, n_filters, image_size=(144, 144, 144), transformer_depth=12):
        super().__init__()
        nd = self._ND
        if not isinstance(image_size, tuple):
            image_size = (image_size,) * nd
        if len(image_size) != nd:
            raise ValueError(f"image_size {image_size} must have {nd} entries")
        self.in_channels, self.n_cls, self.n_filters = in_channels, n_cls, n_filters
        self.image_size, self.transformer_depth = tuple(image_size), transformer_depth
        nf = n_filters
        tokens = 1
        for s in self.image_size:
            tokens *= s // _PATCH
        # registration order == the reference's, so state_dict()/named_parameters() orders agree
        self.attns = nn.ModuleList([_Branch(4 * nf, tokens, transformer_depth // 4, nd) for _ in range(in_channels)])
        self.deep_conv = _ConvUp(4 * nf * in_channels, 8 * nf, nd)
        self.up1 = _ConvUp(8 * nf, 4 * nf, nd)
        self.up2 = _ConvUp(4 * nf, 2 * nf, nd)
        self.up3 = _ConvUp(2 * nf, nf, nd)
        self.block_1_1_left = _ConvNormAct(in_channels, nf, nd)
        self.block_1_2_left = _ConvNormAct(nf, nf, nd)
        self.block_2_1_left = _ConvNormAct(nf, 2 * nf, nd)
        self.block_2_2_left = _ConvNormAct(2 * nf, 2 * nf, nd)
        self.block_3_1_left = _ConvNormAct(2 * nf, 4 * nf, nd)
        self.block_3_2_left = _ConvNormAct(4 * nf, 4 * nf, nd)
        self.block_4_1_left = _ConvNormAct(4 * nf, 8 * nf, nd)
        self.block_4_2_left = _ConvNormAct(8 * nf, 8 * nf, nd)
        self.upconv_3 = _CONVT[nd](8 * nf, 4 * nf, kernel_size=3, stride=2, padding=1, output_padding=1)
        self.block_3_1_right = _ConvNormAct(8 * nf, 4 * nf, nd)
        self.block_3_2_right = _ConvNormAct(4 * nf, 4 * nf, nd)
        self.upconv_2 = _CONVT[nd](4 * nf, 2 * nf, kernel_size=3, stride=2, padding=1, output_padding=1)
        self.block_2_1_right = _ConvNormAct(4 * nf, 2 * nf, nd)
        self.block_2_2_right = _ConvNormAct(2 * nf, 2 * nf, nd)
        self.upconv_1 = _CONVT[nd](2 * nf, nf, kernel_size=3, stride=2, padding=1, output_padding=1)
        self.block_1_1_right = _ConvNormAct(2 * nf, nf, nd)
        self.block_1_2_right = _ConvNormAct(nf, nf, nd)
        self.conv1x1 = _CONV[nd](nf, n_cls, kernel_size=1)
        self.conv1x1_d1 = _CONV[nd](2 * nf, n_cls, kernel_size=1)
        self.conv1x1_d2 = _CONV[nd](4 * nf, n_cls, kernel_size=1)
        self.conv1x1_d3 = _CONV[nd](8 * nf, n_cls, kernel_size=1)

        self._params_checked = None      # parameter list verified by the last forward (see forward)
        self._any_requires_grad = True
        self.compute_dtype = None        # None: follow autocast; "fp32" / "bf16" / "fp16": force
        self.dropout_seed = 0            # base seed of the counter-hash dropout masks (train mode)
        self._step = 0
        self._forced_seed = None
        self._offsets = []
        self._flat = None                # flat fp32 parameter buffer (param.data are views of it)
        self._flat_grad = None
        self._plans, self._runtimes = {}, {}
        self.grad_hook = None            # callable(stage:int) used by hdf_rt.parallel for comm/compute overlap

    def _replicate_for_data_parallel(self):
        """nn.DataParallel(net) with more than one device (trainer.py:228-229) replicates the module per step inside
        one process; this build is one process per GPU (the flat parameter buffer, the plan's workspace and its
        streams belong to one device).  With a single device DataParallel never replicates and just calls forward."""
        raise _lib.HdfError(
            "nn.DataParallel over several GPUs is not supported by the MI355X build: launch one process per GPU "
            "(python -m torch.distributed.run --nproc-per-node N ...), init_process_group('nccl') and attach "
            "hdf_rt.parallel.GradSync(net) as net.grad_hook (INTEGRATION.md section 4); "
            "DataParallel(net, device_ids=[one device]) works")

    # -------------------------------------------------------------- flat parameter management
    def _plan(self, dtype):
        if dtype not in self._plans:
            self._plans[dtype] = Plan(self.in_channels, self.n_cls, self.n_filters, self.image_size,
                                      self.transformer_depth, dtype, embedded_2d=getattr(self, "_embedded_2d", False))
        return self._plans[dtype]

    def _walk_params(self):
        """Parameters in registration order (== self.parameters(), which spends 2.3 ms per call on name building
        for the 1 420 tensors; this walk takes 0.5 ms and runs once per forward)."""
        out = []

        def walk(mod):
            for p in mod._parameters.values():
                if p is not None:
                    out.append(p)
            for c in mod._modules.values():
                if c is not None:
                    walk(c)
        walk(self)
        return out

    def _aliased(self, params=None):
        """True when EVERY parameter is still the fp32 view of the flat buffer it was given by _flatten: a replaced
        tensor (`net.conv1x1 = nn.Conv3d(..)`, `p.data = ..`, load_state_dict(assign=True)) must not leave the
        kernels reading a stale copy."""
        if self._flat is None:
            return False
        base, dev = self._flat.data_ptr(), self._flat.device
        params = self._walk_params() if params is None else params
        offs = self._offsets
        if len(params) != len(offs):
            return False
        f32 = torch.float32
        for p, off in zip(params, offs):
            if p.data_ptr() != base + off or p.dtype != f32 or p.device != dev:
                return False
        return True

    def _flatten(self):
        """(Re)build the flat buffer from the current parameters and re-point every param.data at it."""
        plan = self._plan(_lib.F32)
        named = list(self.named_parameters())
        if [n for n, _ in named] != [t[0] for t in plan.table]:
            raise _lib.HdfError("parameter names/order differ from the plan's state_dict table")
        dev = named[0][1].device
        flat = torch.zeros(plan.param_floats, dtype=torch.float32, device=dev)
        for (name, p), (_, off, numel, shape) in zip(named, plan.table):
            if tuple(p.shape) != tuple(shape):
                raise _lib.HdfError(f"{name}: shape {tuple(p.shape)} != plan {shape}")
            if p.dtype != torch.float32:
                raise _lib.HdfError(f"{name}: master parameters must stay fp32 (got {p.dtype}); use autocast or "
                                    f"compute_dtype='bf16' for low-precision compute")
            view = flat[off: off + numel].view(shape)
            view.copy_(p.data)
            p.data = view
            p.grad = None
        self._flat, self._flat_grad = flat, None
        self._offsets = [4 * t[1] for t in plan.table]
        self._grad_views = None
        self._runtimes = {}

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._flat = None                 # .cuda()/.to()/.float() replaced param.data: re-flatten lazily
        self._params_checked = None
        return out

    def checked_parameters(self):
        """The parameter list the last forward verified (registration order), or a fresh walk: what an optimizer
        that runs once per step can iterate without paying nn.Module.parameters()'s name building (2.3 ms)."""
        return self._params_checked if self._params_checked is not None else self._walk_params()

    def flat_parameters(self, params=None):
        if not self._aliased(params):
            self._flatten()
        return self._flat

    def step_seed(self, step, rank=0):
        """32-bit dropout seed of training forward number `step` on data-parallel rank `rank`."""
        return (self.dropout_seed * 1000003 + step + rank * 0x9E3779B1) & 0xFFFFFFFF

    def set_dropout_seed(self, seed):
        """The next train-mode forward draws its dropout masks from exactly this 32-bit seed (reproducible runs and
        the parity tests against fixtures generated with a given seed); later forwards continue from the counter."""
        self._forced_seed = int(seed) & 0xFFFFFFFF

    def flat_grads(self):
        if self._flat_grad is None:
            self.flat_parameters()
            self._flat_grad = torch.zeros_like(self._flat)
            tbl = self._plan(_lib.F32).table
            self._grad_views = [self._flat_grad[off: off + numel].view(shape) for (_, off, numel, shape) in tbl]
        return self._flat_grad

    def weight_decay_mask(self):
        """uint8 mask over the flat buffer: 1 where trainer.py:812-817 applies weight decay
        (ndim > 1 and name not ending in '.bias')."""
        flat = self.flat_parameters()
        mask = torch.zeros(flat.numel(), dtype=torch.uint8)
        for name, off, numel, shape in self._plan(_lib.F32).table:
            if not (len(shape) == 1 or name.endswith(".bias")):
                mask[off: off + numel] = 1
        return mask.to(flat.device)

    # ------------------------------------------------------------------------------ execution
    def _pick_dtype(self, x):
        if self.compute_dtype is not None:
            return {"fp32": _lib.F32, "bf16": _lib.BF16, "fp16": _lib.F16}[self.compute_dtype]
        if torch.is_autocast_enabled("cuda"):
            adt = torch.get_autocast_dtype("cuda")
            if adt == torch.bfloat16:
                return _lib.BF16
            if adt == torch.float16:
                return _lib.F16
            raise _lib.HdfError(f"autocast dtype {adt} is not supported by the HIP path (bfloat16 / float16)")
        return _lib.F32

    def forward(self, x):
        if not x.is_cuda:
            raise _lib.HdfError("HDenseFormer (MI355X build) needs a GPU tensor: there is no CPU fallback path")
        if x.dim() != 2 + self._ND or x.shape[1] != self.in_channels or tuple(x.shape[2:]) != self.image_size:
            raise _lib.HdfError(f"input shape {tuple(x.shape)} does not match (B,{self.in_channels},"
                                f"{self.image_size})")
        # Host work before the first launch is exposed whenever the caller synchronises once per step (the reference
        # trainer does: loss.item(), trainer.py:382-400): the walk over the 1 420 parameters and the check that each is
        # still the view of the flat buffer cost ~1.5 ms.  So the launch is OPTIMISTIC: with a flat buffer from an
        # earlier forward the kernels are enqueued first, on the assumption that nothing was replaced and requires_grad
        # is as it was, and the full check runs while the GPU works; if it fails, the buffer is rebuilt and the forward
        # is launched again (the first result is dropped before anyone can see it).
        params = self._params_checked
        optimistic = self._flat is not None and params is not None
        if not optimistic:
            params = self._walk_params()
            self.flat_parameters(params)
            self._params_checked = params
            self._any_requires_grad = any(p.requires_grad for p in params)
        flat = self._flat
        if flat.device != x.device:
            raise _lib.HdfError(f"parameters on {flat.device}, input on {x.device}")
        dtype = self._pick_dtype(x)
        key = (dtype, x.device)
        if key not in self._runtimes:
            self._runtimes[key] = Runtime(self._plan(dtype), x.device)
        rt = self._runtimes[key]
        xin = x.detach().float().contiguous()
        need_grad = torch.is_grad_enabled() and self._any_requires_grad
        if self.training:
            self._step += 1
        if self._forced_seed is not None and self.training:
            seed, self._forced_seed = self._forced_seed, None
        else:
            # the reference's replicas draw from per-process RNG streams: fold the data-parallel rank in so that the
            # p = 0.5 masks of different ranks are independent (rank 0 / single process: the plain counter)
            rank = 0
            if torch.distributed.is_available() and torch.distributed.is_initialized():
                rank = torch.distributed.get_rank()
            seed = self.step_seed(self._step, rank)
        anchor = torch.zeros(1, device=x.device, requires_grad=need_grad)
        outs = HDFFunction.apply(xin, anchor, self, rt, self.training, seed)
        if optimistic:  # verify what the launch assumed
            params = self._walk_params()
            any_rg = any(p.requires_grad for p in params)
            stale = not self._aliased(params)
            if stale or any_rg != self._any_requires_grad:
                if stale:
                    self._flatten()
                self._params_checked, self._any_requires_grad = params, any_rg
                need_grad = torch.is_grad_enabled() and any_rg
                anchor = torch.zeros(1, device=x.device, requires_grad=need_grad)
                outs = HDFFunction.apply(xin, anchor, self, rt, self.training, seed)
            else:
                self._params_checked = params
        self._last_rt = rt
        return list(outs)

    def _run_backward(self, rt, x, douts):
        params = self._params_checked        # the tensors the forward of this graph ran on
        if params is None:
            params = self._walk_params()
        if self._flat_grad is None:
            self.flat_grads()
        gflat = self._flat_grad
        # torch semantics: .grad accumulates over backward calls until zero_grad().  The C backward
        # OVERWRITES the flat gradient buffer, so carry existing gradients over explicitly.
        prev = None
        if any(p.grad is not None for p in params):
            if all(p.grad is not None and p.grad.data_ptr() == v.data_ptr() for p, v in zip(params, self._grad_views)):
                prev = gflat.clone()                          # usual case: zero_grad(set_to_none=False)
            else:
                prev = torch.zeros_like(gflat)
                for p, (_, off, numel, shape) in zip(params, self._plan(_lib.F32).table):
                    if p.grad is not None:
                        prev[off: off + numel].view(shape).copy_(p.grad)
        if self.grad_hook is None or prev is not None:
            rt.backward(x, self._flat, douts, gflat, stages=7)
            if prev is not None:
                gflat.add_(prev)
            if self.grad_hook is not None:           # accumulated gradients: reduce after the add, no overlap
                for stage in (1, 2, 3):
                    self.grad_hook(stage)
        elif hasattr(self.grad_hook, "on_bucket_events") and not getattr(self.grad_hook, "staged", False):
            # ONE backward call (the branch-stream fork stays, no host round trip between the stages); the library
            # hands back one event per gradient bucket, recorded where that bucket becomes final, and the hook makes
            # its communication stream wait for them
            if getattr(rt.plan, "is2d", False) or len(rt.plan.cfg[3]) == 2:
                # 2-D plans: every bucket is final when the 2-D gradients are extracted -- one range, the last event
                evs = rt.backward_events(x, self._flat, douts, gflat)
                self.grad_hook.on_bucket_events([evs[-1]] * len(evs), [(0, gflat.numel())] + [(0, 0)] * (len(evs) - 1))
            else:
                self.grad_hook.on_bucket_events(rt.backward_events(x, self._flat, douts, gflat), rt.grad_buckets())
        else:
            # each stage's parameter gradients are final when it returns: their all-reduce overlaps the next stage
            rt.backward(x, self._flat, douts, gflat, stages=1)      # decoder / encoder / heads
            self.grad_hook(1)
            rt.backward(x, self._flat, douts, gflat, stages=2)      # UpConv chain
            self.grad_hook(2)
            rt.backward(x, self._flat, douts, gflat, stages=4)      # transformer branches
            self.grad_hook(3)
        for p, v in zip(params, self._grad_views):
            if p.requires_grad:
                p.grad = v


def HDenseFormer_32(in_channels, n_cls, image_size, transformer_depth):
    return HDenseFormer(in_channels=in_channels, n_cls=n_cls, image_size=image_size, n_filters=32,
                        transformer_depth=transformer_depth)


def HDenseFormer_16(in_channels, n_cls, image_size, transformer_depth):
    return HDenseFormer(in_channels=in_channels, n_cls=n_cls, image_size=image_size, n_filters=16,
                        transformer_depth=transformer_depth)
