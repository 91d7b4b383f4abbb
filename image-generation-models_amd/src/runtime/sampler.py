"""hipGraph-captured reverse-diffusion sampler.

One denoise iteration (UNet forward + posterior update + device-side t decrement) is captured
once into a hipGraph over static buffers and replayed T times: ~150 kernel launches per step
become one graph launch (the eager loop is launch-bound).  Noise for all steps is
drawn from the device Philox generator outside the graph, one step ahead."""
from __future__ import annotations

import torch

from ..ops import functional as K


class GraphSampler:
    def __init__(self, gd, shape):
        self.gd, self.shape = gd, tuple(shape)
        dev = gd.betas.device
        self.x = torch.zeros(shape, device=dev)
        # the same image as NHWC (channel pitch 4, padding zero): what the UNet reads.  The posterior kernel writes both forms, so an
        # iteration has no layout conversion and no copy of its own (round 5: three launches of ~5 us less per step)
        self.xh_full = torch.zeros((shape[0], shape[2], shape[3], (shape[1] + 3) // 4 * 4), device=dev)
        self.xh = self.xh_full[..., :shape[1]]
        self.z = torch.zeros(shape, device=dev)
        self.t = torch.zeros((shape[0],), device=dev, dtype=torch.long)
        self.one = torch.ones((shape[0],), device=dev, dtype=torch.long)
        self.graph = None
        self.tb_table = None          # per-timestep time biases, computed once per capture (weights are frozen while sampling)

    def _iteration(self):
        gd = self.gd
        if K.debug_knob("MI_SAMPLER_INPLACE", "1") != "1":          # round 4's iteration (A/B): a layout conversion in, a copy out
            eps, _ = gd.denoise_fn.forward_nhwc(K.nchw_to_nhwc(self.x), self.t, record=False, time_bias_table=self.tb_table)
            xp, _ = K.p_sample_update(self.x, eps, self.z, self.t, gd._tables(), clip=True, want_nhwc=False)
            self.x.copy_(xp)
            self.t.sub_(self.one)
            return
        eps, _ = gd.denoise_fn.forward_nhwc(self.xh, self.t, record=False, time_bias_table=self.tb_table)
        K.p_sample_update(self.x, eps, self.z, self.t, gd._tables(), clip=True, out=self.x, out_nhwc=self.xh_full)
        self.t.sub_(self.one)

    def set_image(self, x):
        """The ONLY way to put an image into the sampler: the UNet reads the NHWC buffer (`xh`), the posterior kernel the NCHW one
        (`x`, read-only for callers) -- writing `gs.x` directly would leave the two apart and the next step would predict epsilon
        for the old image."""
        self.x.copy_(x)
        self.xh.copy_(K.nchw_to_nhwc(self.x))

    def refresh(self):
        """Bring everything the captured graph reads from static buffers up to date with the current weights: the time-bias
        table, and in bf16 mode the bf16 weight copies (the pack launch runs from Python in forward_nhwc only when the master
        buffer changed, so it is NOT part of the captured iteration -- without this a replay after an optimizer step or
        load_state_dict would convolve with the old weights)."""
        net = self.gd.denoise_fn
        if getattr(net, "compute_mode", "fp32") == "bf16":
            net._shadows()
        tab = net.time_bias_table(self.gd.num_timesteps)
        if self.tb_table is None:
            self.tb_table = tab
        else:
            self.tb_table.copy_(tab)

    def _capture(self):
        self.refresh()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):                      # warm-up outside capture (allocator, lazy init)
            self.t.fill_(1)
            self.set_image(torch.zeros_like(self.x))
            self._iteration()
        torch.cuda.current_stream().wait_stream(s)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self._iteration()

    @torch.no_grad()
    def run(self, record=None):
        gd = self.gd
        if self.graph is None:
            self._capture()
        else:
            self.refresh()
        # noise: device Philox by default; `gd.noise_source` (parity runs) supplies a host tape in the reference's draw order --
        # randn(shape) for x_T, then one draw per step (ddpm.py:404-408,268-273)
        self.set_image(gd._randn(self.shape, self.x.device))
        self.t.fill_(gd.num_timesteps - 1)
        for _ in range(gd.num_timesteps):
            if gd.noise_source is None:
                self.z.normal_()                    # the same Philox draw as randn(shape), straight into the static buffer
            else:
                self.z.copy_(gd._randn(self.shape, self.x.device))
            self.graph.replay()
            if record is not None:
                record.append(self.x.clone())
        return self.x.clone()
