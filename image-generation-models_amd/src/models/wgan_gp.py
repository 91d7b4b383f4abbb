"""`WGAN` (WGAN-GP) with the reference's surface (src/models/wgan_gp.py:12-113) on the HIP kernels.

Same constructor, attribute names (`generator`, `discriminator`), manual optimization with two Adam optimizers, the same
critic / generator alternation (`batch_idx % (n_critic + 1) == n_critic`), the same random draws in the same order from
torch's global CPU generator (z first, then the interpolation weights), the same logged keys (typos included).

The critic step is where this path differs from everything else in the package: the penalty
mean((||d D(x^)/d x^||_2 - 1)^2) (wgan_gp.py:83-97) is a function of an input GRADIENT, so its parameter gradient is a
second derivative.  Nothing here records an autograd graph: `Encoder.input_grad_chain` produces d D/d x^ with the input-gradient
kernels, `mi_gp_penalty` the penalty and its derivative wrt that gradient, and `Encoder.penalty_backward` walks the chain's
adjoint (forward convolutions with the same weights, the norm layers' backward-of-backward, LeakyReLU masks) and then a
standard backward seeded with the terms that reach the forward activations.
"""
import torch

from ..ops import functional as K
from .base import BaseModel, ValidationResult


def _base(t):
    return t._base if t._base is not None else t


class WGAN(BaseModel):
    def __init__(self, datamodule, netG, netD, latent_dim=100, n_critic=5, lrG: float = 1e-4, lrD: float = 1e-4, b1: float = 0,
                 b2: float = 0.9, gp_weight=10):
        super().__init__(datamodule)
        self.save_hyperparameters()
        self.automatic_optimization = False
        try:                                                    # pragma: no cover - hydra is not in this image
            from hydra.utils import instantiate
        except Exception:                                       # noqa: BLE001
            from ..runtime.config import instantiate
        self.generator = instantiate(netG, input_channel=latent_dim, output_channel=self.channels, norm_type="layer")
        self.discriminator = instantiate(netD, input_channel=self.channels, output_channel=1, norm_type="layer")

    def forward(self, z):
        output = self.generator(z)
        return output.reshape(z.shape[0], self.channels, self.height, self.width)

    def flat_nets(self):
        return [self.generator, self.discriminator]

    def configure_optimizers(self):
        from ..runtime.optim import FlatAdam
        hp = self.hparams
        opt_g = FlatAdam(self.generator, lr=hp.lrG, betas=(hp.b1, hp.b2))
        opt_d = FlatAdam(self.discriminator, lr=hp.lrD, betas=(hp.b1, hp.b2))
        return opt_g, opt_d

    def _sync(self, net):
        from ..runtime.ddp import allreduce_flat_grads
        allreduce_flat_grads([net])

    def training_step(self, batch, batch_idx):
        imgs, _ = batch
        G, D, hp = self.generator, self.discriminator, self.hparams
        N = imgs.shape[0]
        z = torch.randn(N, hp.latent_dim).type_as(imgs)                       # CPU draw, like the reference
        opt_g, opt_d = self.optimizers()
        zin = z.float().contiguous().view(N, 1, 1, hp.latent_dim)

        if batch_idx % (hp.n_critic + 1) == hp.n_critic:
            # ---- generator step (wgan_gp.py:64-72): g_loss = -mean D(G(z))
            fake, tape_g = G.forward_nhwc(zin, record=True)
            out, tape_d = D.forward_nhwc(fake, record=True)
            g_loss = -_mean0(out)
            self.log("train_loss/g_loss", g_loss, prog_bar=True)
            dout = torch.zeros_like(_base(out))
            dout[..., 0] = -1.0 / N
            dfake = D.backward_nhwc(tape_d, dout[..., :1], need_dx=True, param_grads=False)
            G.backward_nhwc(tape_g, dfake[..., :self.channels])
            self._sync(G)
            opt_g.step()
            return g_loss

        # ---- critic step (wgan_gp.py:74-107)
        real = _base(K.nchw_to_nhwc(imgs.float()))                            # [N, H, W, 4]: padded lane stays 0
        fake, _ = G.forward_nhwc(zin, record=False)                           # detached in the reference (:79, :85)
        fake = _base(fake)
        lerp = torch.zeros(N, 1, 1, 1).uniform_().to(imgs.device)
        inter = K.lerp_rows(real, fake, lerp.view(N).float().contiguous())
        C = self.channels
        out_rf, tape_rf = D.forward_nhwc(torch.cat([real, fake])[..., :C], record=True)
        out_i, tape_i = D.forward_nhwc(inter[..., :C], record=True)
        real_loss = -_mean0(out_rf[:N])
        fake_loss = _mean0(out_rf[N:])
        g, chain = D.input_grad_chain(tape_i)
        gradient_panelty, u0 = K.gp_penalty(g, scale=float(hp.gp_weight))
        d_loss = real_loss + fake_loss + hp.gp_weight * gradient_panelty
        self.log("train_loss/d_loss", d_loss)
        self.log("train_log/real_logit", -real_loss)
        self.log("train_log/fake_logit", fake_loss)
        self.log("train_log/gradient_panelty", gradient_panelty)
        dout = torch.zeros_like(_base(out_rf))
        dout[:N, ..., 0] = -1.0 / N
        dout[N:, ..., 0] = 1.0 / N
        D.backward_nhwc(tape_rf, dout[..., :1])                               # zeroes, then real + fake terms
        D.penalty_backward(tape_i, chain, u0)                                 # adds gp_weight * d penalty / d theta
        self._sync(D)
        opt_d.step()
        return d_loss

    def validation_step(self, batch, batch_idx):
        img, _ = batch
        z = torch.randn(img.shape[0], self.hparams.latent_dim).to(img.device)
        with torch.no_grad():
            fake_imgs = self.forward(z)
        return ValidationResult(real_image=img, fake_image=fake_imgs)


def _mean0(out):
    """mean over samples of a [N, 1, 1, 1] critic output (a view of a padded buffer)."""
    n = out.shape[0]
    acc = torch.zeros(4, device=out.device)
    K.colsum(out, acc[:out.shape[-1]])
    return acc[0] / n
