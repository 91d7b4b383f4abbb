"""DCGAN-style generator / critic of the reference's conv32 / conv64 networks (src/networks/conv32.py, conv64.py) on the HIP
kernels, restricted to what the WGAN-GP path uses: `norm_type="layer"` (= nn.GroupNorm(1, C), forced by
src/models/wgan_gp.py:30-31), Tanh output, no feature extraction.  Same constructors, parameter names (`main.N.weight`) and
seeded initial weights.

Besides forward / backward the critic implements what the gradient penalty needs (wgan_gp.py:83-107):
  `input_grad_chain`  -- d critic(x) / d x by the input-gradient kernels alone, keeping every intermediate,
  `penalty_backward`  -- the backward of that chain: parameter gradients of a function of the input gradient, including the
                         terms that flow back into the forward activations through the norm layers' backward.
"""
import torch

from ..ops import functional as K
from .flatnet import FlatNet


class _Decoder(FlatNet):
    """z [N, input_channel] -> image [N, C, S, S]: ConvT(k0, 1, 0) -> 3 x ConvT(4, 2, 1) -> ConvT(4, 2, 1) -> Tanh."""
    K0 = 4

    def __init__(self, input_channel=1, output_channel=3, ngf=32, norm_type="batch", output_act="tanh"):
        super().__init__()
        if norm_type != "layer" or output_act != "tanh":
            raise NotImplementedError("only norm_type='layer' + tanh (the WGAN-GP configuration) is built on the HIP path")
        self.input_channel, self.output_channel = input_channel, output_channel
        chans = [input_channel, ngf * 8, ngf * 4, ngf * 2, ngf]
        for i in range(4):
            self._conv_params(f"main.{3 * i}.", chans[i], chans[i + 1], self.K0 if i == 0 else 4, transposed=True)
            self._norm_params(f"main.{3 * i + 1}.", chans[i + 1])
        self._conv_params("main.12.", ngf, output_channel, 4, transposed=True)
        self._finish()

    def _geom(self, i):
        return (self.K0, 1, 0) if i == 0 else (4, 2, 1)

    def forward(self, input):
        return super().forward(input.reshape(input.shape[0], -1, 1, 1))

    def forward_nhwc(self, z, record=False):
        sv = self._sv
        tape = [z] if record else None
        h = z
        for i in range(4):
            k, s, p = self._geom(i)
            a = self._conv(h, f"main.{3 * i}.", k, s, p, transposed=True)
            n, st = K.sample_norm_fwd(a, sv[f"main.{3 * i + 1}.weight"], sv[f"main.{3 * i + 1}.bias"])
            h = K.relu_fwd(n, inplace=True)
            if record:
                tape.append((a, st, h))
        out = self._conv(h, "main.12.", 4, 2, 1, transposed=True)
        y = K.tanh_fwd(out._base if out._base is not None else out, inplace=True)          # padded lanes stay 0
        y = y[..., :self.output_channel]
        if record:
            tape.append(y)
        return y, tape

    def backward_nhwc(self, tape, dy, need_dx=False):
        sv, gv = self._sv, self._begin_backward()
        y = tape[-1]
        yb = y._base if y._base is not None else y
        d = torch.zeros_like(yb)
        d[..., :self.output_channel] = dy
        K.tanh_bwd(yb, d, out=d)
        g = d[..., :self.output_channel]
        for i in range(4, 0, -1):
            a, st, h = tape[i]
            pre = "main.12." if i == 4 else f"main.{3 * i}."
            k, s, p = (4, 2, 1)
            g = self._conv_bwd(g, h, pre, k, s, p, transposed=True)
            K.relu_bwd(h, g, out=g)
            g = K.sample_norm_bwd(a, st, sv[f"main.{3 * i - 2}.weight"], g, dgamma=gv[f"main.{3 * i - 2}.weight"],
                                  dbeta=gv[f"main.{3 * i - 2}.bias"], out=g)
        k, s, p = self._geom(0)
        return self._conv_bwd(g, tape[0], "main.0.", k, s, p, transposed=True, want_dx=need_dx)


class _Encoder(FlatNet):
    """image -> [N, output_channel]: Conv(4,2,1)+LeakyReLU, 3 x (Conv(4,2,1)+GroupNorm(1)+LeakyReLU), Conv(k0,1,0)."""
    K0 = 4
    CONVS = ("main.0.", "main.2.", "main.5.", "main.8.", "main.11.")
    NORMS = (None, "main.3.", "main.6.", "main.9.")

    def __init__(self, input_channel, output_channel, ndf, norm_type="batch", return_features=False):
        super().__init__()
        if norm_type != "layer" or return_features:
            raise NotImplementedError("only norm_type='layer' without feature extraction (the WGAN-GP configuration) is built on the HIP path")
        self.input_channel, self.output_channel = input_channel, output_channel
        chans = [input_channel, ndf, ndf * 2, ndf * 4, ndf * 8]
        for i in range(4):
            self._conv_params(self.CONVS[i], chans[i], chans[i + 1], 4)
            if self.NORMS[i]:
                self._norm_params(self.NORMS[i], chans[i + 1])
        self._conv_params(self.CONVS[4], ndf * 8, output_channel, self.K0)
        self._finish()

    def forward(self, input):
        return super().forward(input).reshape(input.shape[0], -1)

    # ------------------------------------------------------------------ forward / backward
    def forward_nhwc(self, x, record=False):
        sv = self._sv
        tape = [x] if record else None
        h = x
        for i in range(4):
            a = self._conv(h, self.CONVS[i], 4, 2, 1)
            if self.NORMS[i]:
                n, st = K.sample_norm_fwd(a, sv[self.NORMS[i] + "weight"], sv[self.NORMS[i] + "bias"])
            else:
                n, st = a, None
            h = K.leaky_relu_fwd(n, 0.2, inplace=True)
            if record:
                tape.append((a if st is not None else None, st, h))
        out = self._conv(h, self.CONVS[4], self.K0, 1, 0)
        return out, tape

    def backward_nhwc(self, tape, dout, need_dx=False, extra=None, keep_grads=False, param_grads=True):
        """Standard backward.  dout None = zero (then only `extra` drives it); extra[i] (i = 1..3) is added to the gradient of
        the i-th norm layer's input -- the penalty's second-order terms.  keep_grads: do not zero the gradient buffer first.
        param_grads=False: input gradient only (the generator step needs nothing else from the critic)."""
        sv = self._sv
        if not param_grads:
            gv = {}
        elif keep_grads:
            self.flat_grads
            gv = self._gv
        else:
            gv = self._begin_backward()
        g = None
        if dout is not None:
            g = self._conv_bwd(dout, tape[4][2], self.CONVS[4], self.K0, 1, 0, want_dw=param_grads)
        for i in range(3, -1, -1):
            a, st, h = tape[i + 1]
            if g is not None:
                K.leaky_relu_bwd(h, g, 0.2, out=g)
            if st is not None:
                ex = extra[i] if extra is not None else None
                if g is None:
                    g = ex                                                   # GroupNorm backward of a zero gradient is zero
                else:
                    g = K.sample_norm_bwd(a, st, sv[self.NORMS[i] + "weight"], g, dgamma=gv.get(self.NORMS[i] + "weight"),
                                          dbeta=gv.get(self.NORMS[i] + "bias"), extra=ex, out=g)
            if g is None:
                continue
            g = self._conv_bwd(g, tape[i] if i == 0 else tape[i][2], self.CONVS[i], 4, 2, 1, want_dx=(i > 0 or need_dx), want_dw=param_grads)
        return g

    # ------------------------------------------------------------------ gradient penalty (wgan_gp.py:83-107)
    def input_grad_chain(self, tape):
        """d sum(critic(x)) / d x through the input-gradient kernels only (no parameter gradients), every intermediate kept:
        returns (g [N,H,W,4-padded dense], chain) with chain[i] = (d n_i, d a_i) for layer i (d n_0 is None)."""
        sv = self._sv
        x = tape[0]
        N = x.shape[0]
        h4 = tape[4][2]
        ones = torch.zeros((N, 1, 1, 4), device=x.device)
        ones[..., 0] = 1.0
        g = self._conv_bwd(ones[..., :self.output_channel], None, self.CONVS[4], self.K0, 1, 0, want_dw=False, in_hw=(h4.shape[1], h4.shape[2]))
        chain = [None] * 4
        for i in range(3, -1, -1):
            a, st, h = tape[i + 1]
            dn = K.leaky_relu_bwd(h, g, 0.2)
            da = K.sample_norm_bwd(a, st, sv[self.NORMS[i] + "weight"], dn) if st is not None else dn
            chain[i] = (dn if st is not None else None, da)
            src = tape[i] if i == 0 else tape[i][2]
            if i == 0:
                buf = torch.zeros((N, src.shape[1], src.shape[2], (self.input_channel + 3) // 4 * 4), device=x.device)
                self._conv_bwd(da, None, self.CONVS[0], 4, 2, 1, want_dw=False, in_hw=(src.shape[1], src.shape[2]), dx_out=buf[..., :self.input_channel])
                g = buf
            else:
                g = self._conv_bwd(da, None, self.CONVS[i], 4, 2, 1, want_dw=False, in_hw=(src.shape[1], src.shape[2]))
        return g, chain

    def penalty_backward(self, tape, chain, u0):
        """Parameter gradients of a scalar whose gradient wrt the input gradient g is u0 (dense, padded like g); accumulates
        into the flat gradient buffer (call after a backward, or zero the buffer yourself)."""
        sv = self._sv
        self.flat_grads
        gv = self._gv
        mode = self._conv_mode()
        extra = [None] * 4
        w = u0[..., :self.input_channel]
        for i in range(4):
            dn, da = chain[i]
            a, st, h = tape[i + 1]
            self._wgrad_only(w, da, self.CONVS[i], 4, 2, 1)                # d<u, convT(W_i, da_i)>/dW_i
            v = self._conv(w, self.CONVS[i], 4, 2, 1, bias=False)          # adjoint of da_i
            if st is not None:
                t, extra[i] = K.sample_norm_bwd2(a, st, sv[self.NORMS[i] + "weight"], dn, v, dgamma=gv[self.NORMS[i] + "weight"])
            else:
                t = v
            w = K.leaky_relu_bwd(h, t, 0.2, out=t)                         # adjoint of d h_i (mask is piecewise constant)
        N = w.shape[0]
        ones = torch.zeros((N, 1, 1, 4), device=w.device)
        ones[..., 0] = 1.0
        self._wgrad_only(w, ones[..., :self.output_channel], self.CONVS[4], self.K0, 1, 0)
        self.backward_nhwc(tape, None, extra=extra, keep_grads=True)

    def _conv_mode(self):
        from ..models.ddpm import _mode_id
        return _mode_id(self.compute_mode)

    def _wgrad_only(self, inp, dy, pre, k, stride, pad):
        w = self._sv[pre + "weight"]
        kh, kw, ci, co = w.shape
        K.conv_wgrad(inp, dy, self._gv[pre + "weight"], kh=kh, kw=kw, stride=stride, pad=pad, gather_i=True, Ci=ci, Cj=co,
                     grid_g=(inp.shape[1], inp.shape[2]), grid_d=(dy.shape[1], dy.shape[2]), mode=self._conv_mode())
