"""VQ-VAE encoder / decoder with the reference's surface (src/networks/vqvae.py:51-137) on the HIP kernels.

Same constructors (`Encoder(input_channel, output_channel, n_res_layers=3, res_h_dim=128)`,
`Decoder(input_channel, output_channel, h_dim=128, n_res_layers=3, res_h_dim=128)`), same parameter names
(`conv_stack.N...`, `inverse_conv_stack.N...`, `...stack.I.res_block.J.weight`), same seeded initial weights.
Two behaviours of the reference that are easy to miss and are kept:
  * `ResidualStack` repeats ONE `ResidualLayer` object n times (`[layer] * n`, :41-43): the n layers share their
    weights, `state_dict()` lists them under every index, `parameters()` once;
  * `ResidualLayer` opens with `nn.ReLU(True)` (:16), which rectifies x IN PLACE before `x + res_block(x)` (:25):
    the skip connection carries relu(x), not x.
"""
from ..ops import functional as K
from .flatnet import FlatNet


class _ResStackMixin:
    """ResidualStack(in_dim, h_dim, res_h_dim, n) over dense NHWC tensors (vqvae.py:28-49)."""

    def _declare_stack(self, pre, in_dim, h_dim, res_h_dim, n):
        self._conv_params(pre + "stack.0.res_block.1.", in_dim, res_h_dim, 3, bias=False)
        self._conv_params(pre + "stack.0.res_block.3.", res_h_dim, h_dim, 1, bias=False)
        for i in range(1, n):
            self._alias(pre + "stack", str(i), "0")

    def _stack_fwd(self, x, pre, n, tape):
        lay = pre + "stack.0.res_block."
        for _ in range(n):
            r = K.relu_fwd(x, inplace=True)                               # nn.ReLU(True): the skip sees relu(x)
            hr = K.relu_fwd(self._conv(r, lay + "1.", 3, 1, 1, bias=False), inplace=True)
            x = self._conv(hr, lay + "3.", 1, 1, 0, bias=False, residual=r)
            if tape is not None:
                tape.append((r, hr))
        return K.relu_fwd(x, inplace=True)                                 # F.relu after the stack (:48)

    def _stack_bwd(self, dout, out, pre, recs):
        lay = pre + "stack.0.res_block."
        g = K.relu_bwd(out, dout.contiguous())
        for r, hr in reversed(recs):
            dhr = self._conv_bwd(g, hr, lay + "3.", 1, 1, 0, bias=False)
            K.relu_bwd(hr, dhr, out=dhr)
            self._conv_bwd(dhr, r, lay + "1.", 3, 1, 1, bias=False, dx_out=g, accumulate=True)      # g = d relu(x): skip + branch
            K.relu_bwd(r, g, out=g)
        return g


class Encoder(FlatNet, _ResStackMixin):
    def __init__(self, input_channel, output_channel, n_res_layers=3, res_h_dim=128):
        super().__init__()
        self.n_res_layers = n_res_layers
        c = output_channel
        self._conv_params("conv_stack.0.", input_channel, c // 2, 4)
        self._conv_params("conv_stack.2.", c // 2, c, 4)
        self._conv_params("conv_stack.4.", c, c, 3)
        self._declare_stack("conv_stack.5.", c, c, res_h_dim, n_res_layers)
        self._finish()

    def forward_nhwc(self, x, record=False):
        tape = [] if record else None
        a0 = K.relu_fwd(self._conv(x, "conv_stack.0.", 4, 2, 1), inplace=True)
        a1 = K.relu_fwd(self._conv(a0, "conv_stack.2.", 4, 2, 1), inplace=True)
        c2 = self._conv(a1, "conv_stack.4.", 3, 1, 1)
        recs = [] if record else None
        out = self._stack_fwd(c2, "conv_stack.5.", self.n_res_layers, recs)
        if record:
            tape.extend([x, a0, a1, recs, out])
        return out, tape

    def backward_nhwc(self, tape, dy, need_dx=False):
        x, a0, a1, recs, out = tape
        self._begin_backward()
        g = self._stack_bwd(dy, out, "conv_stack.5.", recs)
        da1 = self._conv_bwd(g, a1, "conv_stack.4.", 3, 1, 1)
        K.relu_bwd(a1, da1, out=da1)
        da0 = self._conv_bwd(da1, a0, "conv_stack.2.", 4, 2, 1)
        K.relu_bwd(a0, da0, out=da0)
        return self._conv_bwd(da0, x, "conv_stack.0.", 4, 2, 1, want_dx=need_dx)


class Decoder(FlatNet, _ResStackMixin):
    def __init__(self, input_channel, output_channel, h_dim=128, n_res_layers=3, res_h_dim=128):
        super().__init__()
        self.n_res_layers = n_res_layers
        self._conv_params("inverse_conv_stack.0.", input_channel, h_dim, 3, transposed=True)
        self._declare_stack("inverse_conv_stack.1.", h_dim, h_dim, res_h_dim, n_res_layers)
        self._conv_params("inverse_conv_stack.2.", h_dim, h_dim // 2, 4, transposed=True)
        self._conv_params("inverse_conv_stack.4.", h_dim // 2, output_channel, 4, transposed=True)
        self._finish()

    def forward_nhwc(self, z, record=False):
        tape = [] if record else None
        d0 = self._conv(z, "inverse_conv_stack.0.", 3, 1, 1, transposed=True)
        recs = [] if record else None
        s = self._stack_fwd(d0, "inverse_conv_stack.1.", self.n_res_layers, recs)
        a1 = K.relu_fwd(self._conv(s, "inverse_conv_stack.2.", 4, 2, 1, transposed=True), inplace=True)
        out = self._conv(a1, "inverse_conv_stack.4.", 4, 2, 1, transposed=True)
        if record:
            tape.extend([z, recs, s, a1])
        return out, tape

    def backward_nhwc(self, tape, dy, need_dx=False):
        z, recs, s, a1 = tape
        self._begin_backward()
        da1 = self._conv_bwd(dy, a1, "inverse_conv_stack.4.", 4, 2, 1, transposed=True)
        K.relu_bwd(a1, da1, out=da1)
        ds = self._conv_bwd(da1, s, "inverse_conv_stack.2.", 4, 2, 1, transposed=True)
        g = self._stack_bwd(ds, s, "inverse_conv_stack.1.", recs)
        return self._conv_bwd(g, z, "inverse_conv_stack.0.", 3, 1, 1, transposed=True, want_dx=need_dx)
