"""Flat Adam over the module's flat parameter buffer: one kernel per step (hdf_adam_step), numerically
torch.optim.Adam with the two parameter groups trainer.py:793-840 builds (L2 weight decay on
ndim>1 non-bias tensors, none on the rest)."""
import torch

from ._lib import check, lib, ptr, stream_ptr


class FlatAdam:
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4):
        self.model = model
        # "params" lets torch.cuda.amp.GradScaler.unscale_/step (trainer.py:374-377) walk the gradients like it does
        # for a torch optimizer; the update itself is one kernel over the flat buffer
        self.param_groups = [{"lr": lr, "betas": betas, "eps": eps, "weight_decay": weight_decay,
                              "params": list(model.parameters())}]
        self.state = {}
        self.step_count = 0
        self._state_for = None

    def _state(self):
        # the list the last forward verified: no second walk over the 1 420 parameters per step (1.5 ms of host time)
        flat = self.model.flat_parameters(self.model.checked_parameters())
        if self._state_for is None or self._state_for.data_ptr() != flat.data_ptr():
            # a new flat buffer (module moved / re-flattened): the moments restart, and so does the bias correction
            self.exp_avg = torch.zeros_like(flat)
            self.exp_avg_sq = torch.zeros_like(flat)
            self.mask = self.model.weight_decay_mask()
            self._state_for = flat
            self.step_count = 0
        return flat

    def zero_grad(self, set_to_none=True):
        for p in self.model.checked_parameters():
            p.grad = None

    @torch.no_grad()
    def step(self, grad_scale=1.0):
        flat = self._state()
        g = self.model.flat_grads()
        self.step_count += 1
        pg = self.param_groups[0]
        check(lib().hdf_adam_step(ptr(flat), ptr(g), ptr(self.exp_avg), ptr(self.exp_avg_sq), ptr(self.mask),
                                  flat.numel(), pg["lr"], pg["betas"][0], pg["betas"][1], pg["eps"],
                                  pg["weight_decay"], self.step_count, grad_scale, stream_ptr()), "hdf_adam_step")
