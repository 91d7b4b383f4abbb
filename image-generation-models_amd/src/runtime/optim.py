"""Fused Adam over the UNet's flat parameter buffer (one kernel launch per step).

Update rule = torch.optim.Adam defaults (eps 1e-8, no weight decay, no amsgrad), which is
what the reference's configure_optimizers builds (src/models/ddpm.py:502-512)."""
from __future__ import annotations

import torch

from ..ops import functional as K


class FlatAdam(torch.optim.Optimizer):
    """`net` is one module exposing flat_params / flat_grads / mark_params_dirty, or a list of them (one launch each)."""

    def __init__(self, net, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, grad_scale=1.0, device_state=False):
        # device_state: step count and learning rate live in a 2-word device tensor (mi_adam_step_dev) -- needed when the step is
        # captured in a hipGraph (src/runtime/graphed.py); a scheduler's new lr is copied there at the next eager call of sync_lr()
        self.device_state = device_state
        self._state = None            # float32[2]: word 0 = the step count as uint32 bits, word 1 = lr
        self.nets = list(net) if isinstance(net, (list, tuple)) else [net]
        self.net = self.nets[0]
        super().__init__([p for n in self.nets for p in n.parameters()], dict(lr=lr, betas=betas, eps=eps))
        self.grad_scale = grad_scale
        self._m = None                # one moment buffer per net (lists in both modes, so checkpoints move between them)
        self._v = None
        self._step = 0

    def zero_grad(self, set_to_none: bool = False):
        # backward overwrites the flat gradient buffer (it zeroes it itself); nothing to do.
        pass

    def _moments(self):
        dev = self.nets[0].flat_params.device
        if self._m is None:
            self._m = [torch.zeros_like(n.flat_params) for n in self.nets]
            self._v = [torch.zeros_like(n.flat_params) for n in self.nets]
        elif self._m[0].device != dev:                       # the model moved (or the state was loaded on the host): follow it
            self._m = [m.to(dev) for m in self._m]
            self._v = [v.to(dev) for v in self._v]
        return self._m, self._v

    def _make_state(self, dev):
        st = torch.zeros(2, device=dev, dtype=torch.float32)
        st.view(torch.int32)[0] = int(self._step)
        st[1] = float(self.param_groups[0]["lr"])
        self._state = st

    def device_step_count(self) -> int:
        return int(self._state.view(torch.int32)[0]) if (self.device_state and self._state is not None) else self._step

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        grp = self.param_groups[0]
        ms, vs = self._moments()
        if self.device_state:
            dev = self.nets[0].flat_params.device
            if self._state is None or self._state.device != dev:
                self._make_state(dev)                        # holds the steps taken so far; the tick below counts this one
            self._step += 1
            K.adam_tick(self._state)
            for n, m, v in zip(self.nets, ms, vs):
                K.adam_step_dev(n.flat_params, n.flat_grads, m, v, self._state, grp["betas"][0], grp["betas"][1], grp["eps"], self.grad_scale)
                n.mark_params_dirty()
            return loss
        self._step += 1
        for n, m, v in zip(self.nets, ms, vs):
            K.adam_step(n.flat_params, n.flat_grads, m, v, grp["lr"], grp["betas"][0], grp["betas"][1], grp["eps"], self._step,
                        self.grad_scale)
            n.mark_params_dirty()
        return loss

    def sync_lr(self):
        """Copy the (possibly scheduler-updated) learning rate into the device state; call outside graph capture / replay."""
        if self._state is not None:
            self._state[1] = float(self.param_groups[0]["lr"])

    def state_dict(self):
        step = self.device_step_count()                      # graph replays count on the device, not in self._step
        return {"step": step, "m": self._m, "v": self._v, "param_groups": [
            {k: v for k, v in g.items() if k != "params"} for g in self.param_groups]}

    def load_state_dict(self, sd):
        """Accepts what state_dict() wrote in either mode (m / v as one tensor or a list of per-net tensors)."""
        def as_list(x):
            if x is None:
                return None
            xs = [x] if torch.is_tensor(x) else list(x)
            if len(xs) != len(self.nets):
                raise ValueError(f"optimizer state holds {len(xs)} moment buffers, the optimizer has {len(self.nets)} nets")
            out = []
            for t, n in zip(xs, self.nets):
                if t.numel() != n.flat_params.numel():
                    raise ValueError("optimizer state does not match the flat parameter buffer")
                out.append(t.detach().to(device=n.flat_params.device, dtype=torch.float32).clone().contiguous())
            return out
        new_m, new_v = as_list(sd.get("m")), as_list(sd.get("v"))
        if (new_m is None) != (new_v is None):
            raise ValueError("optimizer state needs both moments or neither")
        self._step = int(sd["step"])
        if new_m is not None and self._m is not None and self._m[0].device == new_m[0].device:
            for dst, src in zip(self._m + self._v, new_m + new_v):      # in place: a captured step graph keeps reading these buffers
                dst.copy_(src)
        elif new_m is None and self._m is not None:
            for dst in self._m + self._v:
                dst.zero_()
        else:
            self._m, self._v = new_m, new_v
        for g, saved in zip(self.param_groups, sd.get("param_groups", [])):
            for k, v in saved.items():
                if k != "params":
                    g[k] = tuple(v) if k == "betas" else v
        if self._state is not None:                          # in place for the same reason
            self._state.view(torch.int32)[0] = self._step
            self._state[1] = float(self.param_groups[0]["lr"])
