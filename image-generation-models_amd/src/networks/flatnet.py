"""Sequential conv networks on the HIP kernels: flat fp32 parameter storage + NHWC tape execution.

The reference's "networks API" is `Cls(input_channel, output_channel, ...)` returning an nn.Module that maps NCHW to
NCHW (src/networks/base.py:16-20).  `FlatNet` gives such a module the machinery the UNet uses: one flat fp32 buffer that
holds every parameter in kernel layout ([kh][kw][Cin][Cout] for both conv flavours), `nn.Parameter`s that are logical
(PyTorch-shaped) views of it registered under the reference's names so `state_dict()` round-trips, a flat gradient buffer
the kernels accumulate into, and forward/backward as explicit NHWC kernel sequences recorded on a tape.
There is no torch fallback: a CPU tensor raises in the first kernel wrapper.
"""
import math
import os

import numpy as np
from typing import Dict, List, Optional, Tuple

import torch
from torch import nn

from ..models.ddpm import _Entry, _Node, _mode_id
from ..ops import functional as K


class _NetFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, anchor, net):
        xin = K.nchw_to_nhwc(x.float())
        record = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        y, tape = net.forward_nhwc(xin, record=record)
        ctx.net, ctx.tape, ctx.need_dx = net, tape, ctx.needs_input_grad[0]
        return K.nhwc_to_nchw(y)

    @staticmethod
    def backward(ctx, dy):
        net, tape = ctx.net, ctx.tape
        ctx.tape = None
        dx = net.backward_nhwc(tape, K.nchw_to_nhwc(dy.float()), need_dx=ctx.need_dx)
        return (K.nhwc_to_nchw(dx) if dx is not None else None), None, None


class FlatNet(nn.Module):
    """Subclasses call `_declare(...)` for each parameter in the reference's construction order, then `_finish()`."""

    # bf16 mode: keep bf16 weight copies (one pack launch per optimizer step) and route convolutions to the ring / halo-tile kernels
    # that need them.  Off for nets whose shapes those kernels do not cover (28x28 MNIST: odd extents) -- there the extra launch
    # only costs (measured: 100.8k -> 90.3k images/s for the VAE).
    BF16_SHADOWS = True

    def __init__(self):
        super().__init__()
        object.__setattr__(self, "_entries", [])
        object.__setattr__(self, "_aliases", [])
        object.__setattr__(self, "_bn", [])
        self.compute_mode = os.environ.get("MI_DDPM_MODE", "fp32")
        self.accumulate_grads = False

    # ------------------------------------------------------------------ declaration
    def _conv_params(self, pre, i, o, k, bias=True, transposed=False):
        shape = (i, o, k, k) if transposed else (o, i, k, k)
        fan = shape[1] * k * k                                   # torch's fan_in for both flavours
        self._entries.append(_Entry(pre + "weight", shape, "convT" if transposed else "conv", "kaiming", fan, 0))
        if bias:
            self._entries.append(_Entry(pre + "bias", (o,), "plain", "ubias", fan, 0))

    def _norm_params(self, pre, c):
        """nn.GroupNorm affine pair (weight = 1, bias = 0: no random draw)."""
        self._entries.append(_Entry(pre + "weight", (c,), "plain", "one", 0, 0))
        self._entries.append(_Entry(pre + "bias", (c,), "plain", "zero", 0, 0))

    def _batchnorm_params(self, pre, c):
        """nn.BatchNorm2d: affine pair in the flat buffer + running_mean / running_var / num_batches_tracked buffers."""
        self._norm_params(pre, c)
        self._bn.append((pre, c))

    def _alias(self, parent: str, name: str, target: str):
        """Register module `parent.target` a second time as `parent.name` (the reference's `[layer] * n`)."""
        self._aliases.append((parent, name, target))

    def _finish(self):
        off = 0
        for e in self._entries:
            off = (off + 63) // 64 * 64
            e.offset = off
            off += e.numel
        flat = torch.zeros((off + 63) // 64 * 64)
        for e in self._entries:                                  # default torch init in construction order => same seeded weights
            if e.init == "kaiming":
                w = torch.empty(e.shape)
                nn.init.kaiming_uniform_(w, a=math.sqrt(5))
                e.logical_view(flat).copy_(w)
            elif e.init == "ubias":
                bound = 1 / math.sqrt(e.fan_in) if e.fan_in > 0 else 0
                e.logical_view(flat).copy_(torch.empty(e.shape).uniform_(-bound, bound))
            elif e.init == "one":
                e.logical_view(flat).fill_(1.0)
        object.__setattr__(self, "_gflat", None)
        object.__setattr__(self, "_dirty", 0)
        object.__setattr__(self, "_anchor", torch.zeros(1, requires_grad=True))
        plist: List[Tuple[_Entry, nn.Parameter]] = []
        for e in self._entries:
            parts = e.key.split(".")
            node = self
            for name in parts[:-1]:
                if name not in node._modules:
                    node.add_module(name, _Node())
                node = node._modules[name]
            p = nn.Parameter(e.logical_view(flat))
            node.register_parameter(parts[-1], p)
            plist.append((e, p))
        for pre, c in self._bn:                               # buffers in torch's order, after the module's parameters
            node = self
            for name in pre.rstrip(".").split("."):
                node = node._modules[name]
            node.register_buffer("running_mean", torch.zeros(c))
            node.register_buffer("running_var", torch.ones(c))
            node.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        for parent, name, target in self._aliases:
            node = self
            for part in parent.split("."):
                node = node._modules[part]
            node.add_module(name, node._modules[target])
        object.__setattr__(self, "_plist", plist)
        self._bind(flat)

    # ------------------------------------------------------------------ storage management (same contract as the UNet)
    def _bind(self, flat: torch.Tensor):
        object.__setattr__(self, "_flat", flat)
        sv = {}
        for e, p in self._plist:
            p.data = e.logical_view(flat)
            sv[e.key] = e.storage_view(flat)
        object.__setattr__(self, "_sv", sv)
        g = self._gflat
        if g is not None and g.device != flat.device:
            object.__setattr__(self, "_gflat", None)
            for _, p in self._plist:
                p.grad = None

    def _apply(self, fn, recurse=True):
        new = fn(self._flat)
        if new.dtype != torch.float32:
            raise RuntimeError("fp32 master weights only; set compute_mode='bf16' for bf16 matrix math")
        if new is not self._flat:
            self._bind(new)
        for mod in self.modules():                            # the few real buffers (batch-norm running statistics)
            for key, buf in mod._buffers.items():
                if buf is not None:
                    mod._buffers[key] = fn(buf)
        return self

    def _buffer(self, key: str) -> torch.Tensor:
        node = self
        parts = key.split(".")
        for name in parts[:-1]:
            node = node._modules[name]
        return node._buffers[parts[-1]]

    def _bn_fwd(self, x, pre, momentum=0.1, eps=1e-5):
        """nn.BatchNorm2d forward in the module's current mode; returns (y, mean, rstd)."""
        sv = self._sv
        training = self.training
        y, mean, rstd = K.batchnorm_fwd(x, sv[pre + "weight"], sv[pre + "bias"], self._buffer(pre + "running_mean"),
                                        self._buffer(pre + "running_var"), momentum, eps, training)
        if training:
            self._buffer(pre + "num_batches_tracked").add_(1)
        return y, mean, rstd

    @property
    def flat_params(self) -> torch.Tensor:
        return self._flat

    @property
    def flat_grads(self) -> torch.Tensor:
        if self._gflat is None:
            g = torch.zeros_like(self._flat)
            object.__setattr__(self, "_gflat", g)
            gv = {}
            for e, p in self._plist:
                p.grad = e.logical_view(g)
                gv[e.key] = e.storage_view(g)
            object.__setattr__(self, "_gv", gv)
        return self._gflat

    def mark_params_dirty(self):
        object.__setattr__(self, "_dirty", self._dirty + 1)

    def zero_grad(self, set_to_none: bool = False):
        if self._gflat is not None:
            self._gflat.zero_()

    def _shadows(self):
        """bf16 copies of every conv weight for the bf16 matrix-core kernels (same scheme as the UNet): wd = master layout
        [tap][Cin][Cout] (input-gradient operand), wf = [tap][Cout][Cin] (forward operand); one pack launch whenever the fp32
        master buffer changed."""
        flat = self._flat
        key = (flat.data_ptr(), flat._version, self._dirty)
        if getattr(self, "_shadow_key", None) != key:
            sh = getattr(self, "_shadow", None)
            if sh is None or sh[0].device != flat.device:
                ents = []
                for e in self._entries:
                    if e.layout in ("conv", "convT"):
                        kh, kw, ci, co = e.storage_view(flat).shape
                        ents.append((e.offset, kh * kw, ci, co))
                table, nent, tile = K.pack_table(ents, flat.device)
                sh = (torch.zeros(flat.numel(), device=flat.device, dtype=torch.bfloat16),
                      torch.zeros(flat.numel(), device=flat.device, dtype=torch.bfloat16), table, nent, tile)
                object.__setattr__(self, "_shadow", sh)
            wd, wf, table, nent, tiles = sh
            K.pack_weights_bf16(table, nent, tiles, flat, wd, wf)
            object.__setattr__(self, "_shadow_key", key)
        return self._shadow[0], self._shadow[1]

    def _off(self, key: str) -> int:
        offs = getattr(self, "_offs", None)
        if offs is None:
            offs = {e.key: e.offset for e in self._entries}
            object.__setattr__(self, "_offs", offs)
        return offs[key]

    # ------------------------------------------------------------------ execution
    def forward(self, x):
        if self._anchor.device != x.device:
            object.__setattr__(self, "_anchor", torch.zeros(1, device=x.device, requires_grad=True))
        anchor = self._anchor if (torch.is_grad_enabled() and self.training) else self._anchor.detach()
        return _NetFunction.apply(x, anchor, self)

    def forward_nhwc(self, x, record=False):
        raise NotImplementedError

    def backward_nhwc(self, tape, dy, need_dx=False):
        raise NotImplementedError

    def _begin_backward(self):
        g = self.flat_grads
        if not self.accumulate_grads:
            g.zero_()                                             # the weight-gradient kernels accumulate
        return self._gv

    # one convolution (either flavour) and its three gradients on the generic implicit-GEMM kernels
    def _conv(self, inp, pre, k, stride=1, pad=0, transposed=False, bias=True, residual=None):
        w = self._sv[pre + "weight"]
        kh, kw, ci, co = w.shape
        ih, iw = inp.shape[1], inp.shape[2]
        if transposed:
            oh, ow = (ih - 1) * stride - 2 * pad + kh, (iw - 1) * stride - 2 * pad + kw
        else:
            oh, ow = (ih + 2 * pad - kh) // stride + 1, (iw + 2 * pad - kw) // stride + 1
        if (not transposed and pad == 0 and (ih, iw) == (kh, kw) and residual is None and inp.is_contiguous()
                and (co == 1 or co % 4 == 0)):
            # the kernel covers the whole input: a plain GEMM [N, kh*kw*ci] x [kh*kw*ci, co] with a long contraction and few rows
            # (the critic's last layer: 8192 -> 1).  One generic-kernel workgroup would walk that contraction alone (510 us);
            # the split-K small GEMM spreads it over the chip.
            n = inp.shape[0]
            out = torch.zeros((n, 1, 1, (co + 3) // 4 * 4), device=inp.device, dtype=torch.float32)
            a = inp.view(n, kh * kw * ci)
            b = w.view(1, kh * kw * ci) if co == 1 else w.view(kh * kw * ci, co)
            y = K.small_gemm(False, co == 1, a, b, bias=self._sv[pre + "bias"] if bias else None, out=out.view(n, -1)[:, :co],
                             accumulate=True, allow_split=True)
            if y is not None:
                return out[..., :co]
        b = self._sv[pre + "bias"] if bias else None
        wb = None
        if self.compute_mode == "bf16" and self.BF16_SHADOWS and ci % 8 == 0:
            _, wf = self._shadows()
            wb = wf[self._off(pre + "weight"):]
            if k in (1, 3) and stride == 1 and not transposed and pad == k // 2:       # the LDS halo-tile kernel
                y = K.conv3x3_bf16w(inp, wb, K=ci, Nc=co, flip=False, ksize=k, bias=b, residual=residual)
                if y is not None:
                    return y
        return K.conv_igemm(inp, w, kh=kh, kw=kw, stride=stride, pad=pad, transposed=transposed, w_kn=True, K=ci, Nc=co,
                            out_hw=(oh, ow), mode=_mode_id(self.compute_mode), bias=b, residual=residual, wb=wb)

    def _conv_bwd(self, dy, inp, pre, k, stride=1, pad=0, transposed=False, bias=True, want_dx=True, dx_out=None, accumulate=False,
                  want_dw=True, in_hw=None):
        """Gradients of y = conv(inp): weight/bias gradients accumulate into the flat gradient buffer (want_dw), the input
        gradient is returned (want_dx).  inp may be None when only the input gradient is wanted (pass in_hw)."""
        w, mode = self._sv[pre + "weight"], _mode_id(self.compute_mode)
        kh, kw, ci, co = w.shape
        ih, iw = in_hw if inp is None else (inp.shape[1], inp.shape[2])
        oh, ow = dy.shape[1], dy.shape[2]
        gv = self._gv if want_dw else None
        if not want_dw:
            pass
        elif transposed:       # dW[tap][ci][co] = sum over input pixels x[j] * dy[scatter(j, tap)]
            K.conv_wgrad(inp, dy, gv[pre + "weight"], kh=kh, kw=kw, stride=stride, pad=pad, gather_i=False, Ci=ci, Cj=co,
                         grid_g=(oh, ow), grid_d=(ih, iw), mode=mode)
            if bias:
                K.colsum(dy, gv[pre + "bias"])
        else:
            K.conv_wgrad(inp, dy, gv[pre + "weight"], kh=kh, kw=kw, stride=stride, pad=pad, gather_i=True, Ci=ci, Cj=co,
                         grid_g=(ih, iw), grid_d=(oh, ow), mode=mode, dbias=gv[pre + "bias"] if bias else None)
        if not want_dx:
            return None
        wb = None
        if self.compute_mode == "bf16" and self.BF16_SHADOWS and co % 8 == 0:
            wd, _ = self._shadows()
            wb = wd[self._off(pre + "weight"):]
            if k in (1, 3) and stride == 1 and not transposed and pad == k // 2 and kh == k:
                y = K.conv3x3_bf16w(dy, wb, K=co, Nc=ci, flip=True, ksize=k, out=dx_out, accumulate=accumulate)
                if y is not None:
                    return y
        return K.conv_igemm(dy, w, kh=kh, kw=kw, stride=stride, pad=pad, transposed=not transposed, w_kn=False, K=co, Nc=ci,
                            out_hw=(ih, iw), mode=mode, out=dx_out, accumulate=accumulate, wb=wb)
