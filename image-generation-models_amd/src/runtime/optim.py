"""Fused Adam over the UNet's flat parameter buffer (one kernel launch per step).

Update rule = torch.optim.Adam defaults (eps 1e-8, no weight decay, no amsgrad), which is
what the reference's configure_optimizers builds (src/models/ddpm.py:502-512)."""
from __future__ import annotations

import torch

from ..ops import functional as K


class FlatAdam(torch.optim.Optimizer):
    """`net` is one module exposing flat_params / flat_grads / mark_params_dirty, or a list of them (one launch each)."""

    def __init__(self, net, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, grad_scale=1.0, device_state=False):
        # device_state: step count and learning rate live in a 2-float device tensor (mi_adam_step_dev) -- needed when the step is
        # captured in a hipGraph (src/runtime/graphed.py); a scheduler's new lr is copied there at the next eager call of sync_lr()
        self.device_state = device_state
        self._state = None
        self.nets = list(net) if isinstance(net, (list, tuple)) else [net]
        self.net = self.nets[0]
        super().__init__([p for n in self.nets for p in n.parameters()], dict(lr=lr, betas=betas, eps=eps))
        self.grad_scale = grad_scale
        self._m = None
        self._v = None
        self._step = 0

    def zero_grad(self, set_to_none: bool = False):
        # backward overwrites the flat gradient buffer (it zeroes it itself); nothing to do.
        pass

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        grp = self.param_groups[0]
        self._step += 1
        if self.device_state:
            dev = self.nets[0].flat_params.device
            if self._state is None or self._state.device != dev:
                self._state = torch.tensor([float(self._step - 1), float(grp["lr"])], device=dev)
                self._m = [torch.zeros_like(n.flat_params) for n in self.nets]
                self._v = [torch.zeros_like(n.flat_params) for n in self.nets]
            K.adam_tick(self._state)
            for n, m, v in zip(self.nets, self._m, self._v):
                K.adam_step_dev(n.flat_params, n.flat_grads, m, v, self._state, grp["betas"][0], grp["betas"][1], grp["eps"], self.grad_scale)
                n.mark_params_dirty()
            return loss
        if len(self.nets) == 1:
            p, g = self.net.flat_params, self.net.flat_grads
            if self._m is None or self._m.device != p.device:
                self._m = torch.zeros_like(p)
                self._v = torch.zeros_like(p)
            K.adam_step(p, g, self._m, self._v, grp["lr"], grp["betas"][0], grp["betas"][1], grp["eps"], self._step,
                        self.grad_scale)
            self.net.mark_params_dirty()
            return loss
        if self._m is None or self._m[0].device != self.nets[0].flat_params.device:
            self._m = [torch.zeros_like(n.flat_params) for n in self.nets]
            self._v = [torch.zeros_like(n.flat_params) for n in self.nets]
        for n, m, v in zip(self.nets, self._m, self._v):
            K.adam_step(n.flat_params, n.flat_grads, m, v, grp["lr"], grp["betas"][0], grp["betas"][1], grp["eps"], self._step,
                        self.grad_scale)
            n.mark_params_dirty()
        return loss

    def sync_lr(self):
        """Copy the (possibly scheduler-updated) learning rate into the device state; call outside graph capture / replay."""
        if self._state is not None:
            self._state[1] = float(self.param_groups[0]["lr"])

    def state_dict(self):
        step = int(round(float(self._state[0]))) if self._state is not None else self._step     # graph replays count on the device
        return {"step": step, "m": self._m, "v": self._v, "param_groups": [
            {k: v for k, v in g.items() if k != "params"} for g in self.param_groups]}

    def load_state_dict(self, sd):
        self._step, self._m, self._v = sd["step"], sd["m"], sd["v"]
