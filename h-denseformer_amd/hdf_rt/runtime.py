"""Host-side runtime of the HIP path: plan handle, flat parameter / gradient buffers, workspace,
and the torch.autograd.Function that puts HDenseFormer.forward (reference models/HDenseFormer.py:229-255)
and its backward behind ONE autograd node.  PyTorch is used for device memory, streams and autograd
plumbing only; all arithmetic happens inside libhdf_hip.so."""
import ctypes as C

import torch

from . import _lib
from ._lib import BF16, F16, F32, check, lib, ptr, stream_ptr

_TORCH_DTYPE = {F32: torch.float32, BF16: torch.bfloat16, F16: torch.float16}


class Plan:
    """Owns an hdf_plan* (parameter table, workspace layout)."""

    def __init__(self, in_channels, n_cls, n_filters, image_size, transformer_depth, dtype, embedded_2d=False):
        """embedded_2d (2-D plans only): the depth-16 replicated 3-D embedding of rounds 3-5 instead of the native depth-1
        path -- the oracle of tests/test_gpu_model_2d.py"""
        self.cfg = (int(in_channels), int(n_cls), int(n_filters), tuple(int(v) for v in image_size),
                    int(transformer_depth))
        self.dtype = dtype
        h = C.c_void_p()
        if len(self.cfg[3]) == 2:      # HDenseFormer_2D: the plan's parameter table is the 2-D state_dict
            hh, w = self.cfg[3]
            create = lib().hdf_plan_create_2d_embedded if embedded_2d else lib().hdf_plan_create_2d
            check(create(self.cfg[0], self.cfg[1], self.cfg[2], hh, w, self.cfg[4], dtype, C.byref(h)), "hdf_plan_create_2d")
        else:
            d, hh, w = self.cfg[3]
            check(lib().hdf_plan_create(self.cfg[0], self.cfg[1], self.cfg[2], d, hh, w, self.cfg[4], dtype,
                                        C.byref(h)), "hdf_plan_create")
        self.h = h
        self.param_floats = lib().hdf_plan_param_floats(h)
        self.table = []          # (name, offset, numel, shape)
        name = C.create_string_buffer(256)
        off, numel, ndim = C.c_int64(), C.c_int64(), C.c_int()
        shape = (C.c_int64 * 5)()
        for i in range(lib().hdf_plan_num_params(h)):
            check(lib().hdf_plan_param_info(h, i, name, 256, C.byref(off), C.byref(numel), C.byref(ndim), shape),
                  "hdf_plan_param_info")
            self.table.append((name.value.decode(), off.value, numel.value, tuple(shape[k] for k in range(ndim.value))))

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().hdf_plan_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def workspace_bytes(self, batch, backward=True):
        """bytes of the arena for a training step, or (backward=False) of the prefix a forward alone touches"""
        if backward:
            return lib().hdf_plan_workspace_bytes(self.h, batch)
        return lib().hdf_plan_inference_workspace_bytes(self.h, batch)

    def buffer_info(self, batch, name):
        off, pitch = C.c_int64(), C.c_int64()
        c, d, h, w = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        check(lib().hdf_plan_buffer_info(self.h, batch, name.encode(), C.byref(off), C.byref(pitch), C.byref(c),
                                         C.byref(d), C.byref(h), C.byref(w)), "hdf_plan_buffer_info")
        return off.value, pitch.value, c.value, (d.value, h.value, w.value)


    def region_info(self, batch, name):
        """(byte offset, bytes) of a raw fp32 region of the transformer branches inside the workspace (debug / tests)"""
        off, nb = C.c_int64(), C.c_int64()
        check(lib().hdf_plan_region_info(self.h, batch, name.encode(), C.byref(off), C.byref(nb)),
              "hdf_plan_region_info")
        return off.value, nb.value


class Runtime:
    """Per (module, dtype) execution state: plan + workspace.  The flat parameter and gradient buffers
    are owned by the module (shared between dtypes)."""

    def __init__(self, plan, device):
        self.plan = plan
        self.device = device
        self.ws = None
        self.ws_batch = -1
        self.ws_backward = False    # the arena includes the backward scratch
        self.gen = 0            # forward generation; backward must match (single forward in flight)
        self.out_shapes = None

    def _ensure_ws(self, batch, backward):
        """Inference (no autograd graph) gets the forward prefix only: less than half of the training arena."""
        if self.ws is None or self.ws_batch != batch or (backward and not self.ws_backward):
            self.ws = None
            self.ws = torch.empty(self.plan.workspace_bytes(batch, backward), dtype=torch.uint8, device=self.device)
            self.ws_batch, self.ws_backward = batch, backward

    def forward(self, x, flat_params, training, seed, need_backward=True):
        cfg = self.plan.cfg
        b = x.shape[0]
        self._ensure_ws(b, need_backward)
        tdt = _TORCH_DTYPE[self.plan.dtype]
        outs = [torch.empty((b, cfg[1]) + tuple(s >> i for s in cfg[3]), dtype=tdt, device=self.device)
                for i in range(4)]
        check(lib().hdf_forward(self.plan.h, ptr(x), ptr(flat_params), ptr(self.ws), self.ws.numel(),
                                ptr(outs[0]), ptr(outs[1]), ptr(outs[2]), ptr(outs[3]), b, int(bool(training)),
                                int(seed) & 0xFFFFFFFFFFFFFFFF, stream_ptr()), "hdf_forward")
        self.gen += 1
        return outs

    def backward(self, x, flat_params, douts, flat_grads, stages=7):
        b = x.shape[0]
        check(lib().hdf_backward_stages(self.plan.h, ptr(x), ptr(flat_params), ptr(self.ws), self.ws.numel(),
                                        ptr(douts[0]), ptr(douts[1]), ptr(douts[2]), ptr(douts[3]), ptr(flat_grads),
                                        b, stages, stream_ptr()), "hdf_backward")

    NUM_GRAD_BUCKETS = 5        # HDF_NUM_GRAD_BUCKETS (include/hdf.h)

    def grad_buckets(self):
        """[(lo, hi)] float ranges of the gradient buckets of hdf_backward_events (hdf_plan_grad_bucket): 0 decoder +
        heads, 1 UpConv chain, 2 transformer branches, 3 encoder levels 1-3, 4 encoder level 0."""
        out = []
        for k in range(self.NUM_GRAD_BUCKETS):
            lo, hi = C.c_int64(), C.c_int64()
            check(lib().hdf_plan_grad_bucket(self.plan.h, k, C.byref(lo), C.byref(hi)), "hdf_plan_grad_bucket")
            out.append((lo.value, hi.value))
        return out

    def backward_events(self, x, flat_params, douts, flat_grads):
        """The one-call backward; returns NUM_GRAD_BUCKETS opaque event handles: gradient bucket k (grad_buckets()[k]) is
        final once handle k has fired (include/hdf.h: hdf_backward_events)."""
        import ctypes
        b = x.shape[0]
        evs = (ctypes.c_void_p * self.NUM_GRAD_BUCKETS)()
        check(lib().hdf_backward_events(self.plan.h, ptr(x), ptr(flat_params), ptr(self.ws), self.ws.numel(),
                                        ptr(douts[0]), ptr(douts[1]), ptr(douts[2]), ptr(douts[3]), ptr(flat_grads),
                                        b, stream_ptr(), evs), "hdf_backward_events")
        return [evs[k] for k in range(self.NUM_GRAD_BUCKETS)]

    def read_buffer(self, name):
        """Debug / parity helper: copy a named channels-last activation out of the workspace as NCDHW fp32
        (2-D plans: the buffers of the depth-replicated 3-D embedding)."""
        off, pitch, c, (d, h, w) = self.plan.buffer_info(self.ws_batch, name)
        esz = 4 if self.plan.dtype == F32 else 2
        tdt = _TORCH_DTYPE[self.plan.dtype]
        n = self.ws_batch * d * h * w
        # a channel sub-view of a wider buffer starts at `off`; rows are `pitch` elements apart
        raw = self.ws[off: off + ((n - 1) * pitch + c) * esz].view(tdt)
        v = torch.as_strided(raw, (n, c), (pitch, 1))
        return v.reshape(self.ws_batch, d, h, w, c).permute(0, 4, 1, 2, 3).float().contiguous()


    def read_region(self, name):
        """a raw fp32 region of the transformer branches (Plan.region_info) as a flat float32 view of the workspace"""
        off, nb = self.plan.region_info(self.ws_batch, name)
        return self.ws[off: off + nb].view(torch.float32)


class HDFFunction(torch.autograd.Function):
    """One autograd node for the whole network.  Parameter gradients are written straight into the
    module's flat gradient buffer (param.grad are views of it), like fused-optimizer frameworks do;
    the node's own inputs are only x and a dummy anchor that keeps the node alive in the graph."""

    @staticmethod
    def forward(ctx, x, anchor, module, rt, training, seed):
        outs = rt.forward(x, module._flat, training, seed, need_backward=bool(anchor.requires_grad))
        ctx.module, ctx.rt, ctx.gen = module, rt, rt.gen
        ctx.save_for_backward(x)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        module, rt = ctx.module, ctx.rt
        if ctx.gen != rt.gen:
            raise _lib.HdfError("HDenseFormer backward called after another forward of the same module overwrote "
                                "its workspace (one forward in flight per module and dtype)")
        (x,) = ctx.saved_tensors
        tdt = _TORCH_DTYPE[rt.plan.dtype]
        b = x.shape[0]
        cfg = rt.plan.cfg
        douts = []
        for i, g in enumerate(gouts):
            if g is None:
                g = torch.zeros((b, cfg[1]) + tuple(s >> i for s in cfg[3]), dtype=tdt, device=x.device)
            douts.append(g.to(tdt).contiguous())
        module._run_backward(rt, x, douts)
        return None, None, None, None, None, None
