"""MNIST-sized conv encoder / decoder with the reference's surface (src/networks/basic.py:147-204: `ConvEncoder`, `ConvDecoder`,
what configs/networks/conv_mnist.yaml instantiates) on the HIP kernels.  `norm_type` "batch" (the config default:
nn.BatchNorm2d with running statistics) and "layer" (nn.GroupNorm(1, C)) are built; same constructors, parameter / buffer names
(`network.N.weight`, `network.N.running_mean`, ...) and seeded initial weights.  The MLP networks of that file are not on any
BASELINE config's path.
"""
import torch

from ..ops import functional as K
from .flatnet import FlatNet


class _NormMixin:
    def _declare_norm(self, pre, c):
        if self.norm_type == "batch":
            self._batchnorm_params(pre, c)
        else:
            self._norm_params(pre, c)

    def _norm_fwd(self, a, pre):
        if self.norm_type == "batch":
            y, mean, rstd = self._bn_fwd(a, pre)
            return y, (mean, rstd)
        y, st = K.sample_norm_fwd(a, self._sv[pre + "weight"], self._sv[pre + "bias"])
        return y, st

    def _norm_bwd(self, a, st, pre, g, gv):
        if self.norm_type == "batch":
            return K.batchnorm_bwd(a, st[0], st[1], self._sv[pre + "weight"], g, dgamma=gv[pre + "weight"], dbeta=gv[pre + "bias"], out=g)
        return K.sample_norm_bwd(a, st, self._sv[pre + "weight"], g, dgamma=gv[pre + "weight"], dbeta=gv[pre + "bias"], out=g)


def _check_norm(norm_type):
    if norm_type not in ("batch", "layer"):
        raise NotImplementedError(f"norm_type={norm_type!r}: only 'batch' and 'layer' are built on the HIP path")


class ConvDecoder(FlatNet, _NormMixin):
    """z [N, C_in] -> [N, C_out, 28, 28]: ConvT(4,1,0) -> ConvT(3,2,1) -> ConvT(4,2,1) -> ConvT(4,2,1), norm + ReLU between, output act."""
    GEOM = ((4, 1, 0), (3, 2, 1), (4, 2, 1), (4, 2, 1))
    BF16_SHADOWS = False

    def __init__(self, input_channel, output_channel, ngf, norm_type="batch", output_act="tanh"):
        super().__init__()
        _check_norm(norm_type)
        if output_act not in ("tanh", "identity"):
            raise NotImplementedError("output_act: 'tanh' (normalised inputs) or 'identity'")
        self.input_channel, self.output_channel, self.norm_type, self.output_act = input_channel, output_channel, norm_type, output_act
        chans = [input_channel, ngf * 4, ngf * 2, ngf, output_channel]
        for i in range(4):
            self._conv_params(f"network.{3 * i}.", chans[i], chans[i + 1], self.GEOM[i][0], transposed=True)
            if i < 3:
                self._declare_norm(f"network.{3 * i + 1}.", chans[i + 1])
        self._finish()

    def forward(self, x):
        return super().forward(x.reshape(x.shape[0], -1, 1, 1))

    def forward_nhwc(self, z, record=False):
        tape = [z] if record else None
        h = z
        for i in range(3):
            k, s, p = self.GEOM[i]
            a = self._conv(h, f"network.{3 * i}.", k, s, p, transposed=True)
            n, st = self._norm_fwd(a, f"network.{3 * i + 1}.")
            h = K.relu_fwd(n, inplace=True)
            if record:
                tape.append((a, st, h))
        k, s, p = self.GEOM[3]
        out = self._conv(h, "network.9.", k, s, p, transposed=True)
        if self.output_act == "tanh":
            base = out._base if out._base is not None else out
            K.tanh_fwd(base, inplace=True)                                   # padded lanes stay 0
        if record:
            tape.append(out)
        return out, tape

    def backward_nhwc(self, tape, dy, need_dx=False):
        gv = self._begin_backward()
        y = tape[-1]
        yb = y._base if y._base is not None else y
        d = torch.zeros_like(yb)
        d[..., :self.output_channel] = dy
        if self.output_act == "tanh":
            K.tanh_bwd(yb, d, out=d)
        g = d[..., :self.output_channel]
        for i in range(3, 0, -1):
            a, st, h = tape[i]
            k, s, p = self.GEOM[i]
            g = self._conv_bwd(g, h, f"network.{3 * i}.", k, s, p, transposed=True)
            K.relu_bwd(h, g, out=g)
            g = self._norm_bwd(a, st, f"network.{3 * i - 2}.", g, gv)
        k, s, p = self.GEOM[0]
        return self._conv_bwd(g, tape[0], "network.0.", k, s, p, transposed=True, want_dx=need_dx)


class ConvEncoder(FlatNet, _NormMixin):
    """[N, C_in, 28, 28] -> [N, C_out]: Conv(4,2,1)+LeakyReLU, Conv(4,2,1)+norm+LeakyReLU, Conv(3,2,1)+norm+LeakyReLU, Conv(4,1,0)."""
    CONVS = (("network.0.", 4, 2, 1, None), ("network.2.", 4, 2, 1, "network.3."), ("network.5.", 3, 2, 1, "network.6."))
    BF16_SHADOWS = False

    def __init__(self, input_channel, output_channel, ndf, norm_type="batch", return_features=False):
        super().__init__()
        _check_norm(norm_type)
        if return_features:
            raise NotImplementedError("feature extraction is used by the GAN zoo only")
        self.input_channel, self.output_channel, self.norm_type = input_channel, output_channel, norm_type
        chans = [input_channel, ndf, ndf * 2, ndf * 4]
        for i, (pre, k, s, p, norm) in enumerate(self.CONVS):
            self._conv_params(pre, chans[i], chans[i + 1], k)
            if norm:
                self._declare_norm(norm, chans[i + 1])
        self._conv_params("network.8.", ndf * 4, output_channel, 4)
        self._finish()

    def forward(self, x):
        return super().forward(x).reshape(-1, self.output_channel)

    def forward_nhwc(self, x, record=False):
        tape = [x] if record else None
        h = x
        for pre, k, s, p, norm in self.CONVS:
            a = self._conv(h, pre, k, s, p)
            if norm:
                n, st = self._norm_fwd(a, norm)
            else:
                n, st = a, None
            h = K.leaky_relu_fwd(n, 0.2, inplace=True)
            if record:
                tape.append((a if st is not None else None, st, h))
        out = self._conv(h, "network.8.", 4, 1, 0)
        return out, tape

    def backward_nhwc(self, tape, dout, need_dx=False):
        gv = self._begin_backward()
        g = self._conv_bwd(dout, tape[3][2], "network.8.", 4, 1, 0)
        for i in range(2, -1, -1):
            pre, k, s, p, norm = self.CONVS[i]
            a, st, h = tape[i + 1]
            K.leaky_relu_bwd(h, g, 0.2, out=g)
            if st is not None:
                g = self._norm_bwd(a, st, norm, g, gv)
            g = self._conv_bwd(g, tape[0] if i == 0 else tape[i][2], pre, k, s, p, want_dx=(i > 0 or need_dx))
        return g
