"""`VAE` with the reference's surface (src/models/vae.py:11-89) on the HIP kernels -- BASELINE cfg 1 (`experiment=vae/mnist_conv`).

Same constructor (including the `decoder_dist="guassian"` default that the reference's own `get_decode_dist` rejects -- the
configs pass "gaussian"), attribute names, logged keys, `forward(z)` = decode, `sample(N)`, Adam + StepLR(1, 0.99).
`training_step` is one autograd node: encoder -> [mu | log_sigma] -> z = mu + exp(log_sigma) eps -> decoder -> Gaussian
log-likelihood with unit variance (src/utils/distributions.py:16-21) and the KL term (src/utils/losses.py:30-32); eps is drawn
with `torch.randn` on the input's device right where `Normal.rsample()` draws it.
The reference runs this config on the CPU (`trainer=cpu`); here it runs on the GPU like everything else -- a CPU tensor raises.
"""
import math

import torch

from ..ops import functional as K
from .base import BaseModel, ValidationResult


class _VAEStep(torch.autograd.Function):
    @staticmethod
    def forward(ctx, imgs, eps, anchor, model):
        enc, dec, hp = model.encoder, model.decoder, model.hparams
        record = ctx.needs_input_grad[2]
        imgs = imgs.float().contiguous()
        n = imgs.shape[0]
        h, tape_e = enc.forward_nhwc(K.nchw_to_nhwc(imgs), record=record)              # [N,1,1,2L]
        hrows = h.reshape(n, -1)
        z, kld = K.vae_latent_fwd(hrows, eps)
        out, tape_d = dec.forward_nhwc(z.view(n, 1, 1, -1), record=record)
        chw = imgs[0].numel()
        mse, dout = K.eps_loss(out, imgs, 1, want_grad=record, gscale=0.5 * float(hp.recon_weight) * chw)
        log_p = -0.5 * chw * mse - 0.5 * chw * math.log(2 * math.pi)                   # mean over the batch of sum log N(x; recon, 1)
        elbo = -float(hp.beta) * kld + float(hp.recon_weight) * log_p
        ctx.model, ctx.tapes, ctx.saved = model, (tape_e, tape_d), (hrows, eps, dout)
        ctx.mark_non_differentiable(kld, log_p)
        return -elbo, kld, log_p, z, K.nhwc_to_nchw(out)

    @staticmethod
    def backward(ctx, dloss, *_):
        model = ctx.model
        tape_e, tape_d = ctx.tapes
        hrows, eps, dout = ctx.saved
        ctx.tapes = ctx.saved = None
        K.scale_by_device_scalar(dout, dloss)
        dz = model.decoder.backward_nhwc(tape_d, dout, need_dx=True)
        n = hrows.shape[0]
        dh = K.vae_latent_bwd(hrows, eps, dz.reshape(n, -1), float(model.hparams.beta), g_dev=dloss.reshape(1).float())
        model.encoder.backward_nhwc(tape_e, dh.view(n, 1, 1, -1), need_dx=False)
        return None, None, None, None


class _GaussianDistribution:
    """Unit-variance Gaussian decoder (src/utils/distributions.py:12-24): `sample` returns the mean."""

    @staticmethod
    def sample(pred):
        return pred


class VAE(BaseModel):
    def __init__(self, datamodule=None, encoder=None, decoder=None, latent_dim: int = 100, beta: float = 1.0, recon_weight: float = 1.0,
                 lr: float = 1e-4, b1: float = 0.9, b2: float = 0.999, decoder_dist="guassian"):
        super().__init__(datamodule)
        self.save_hyperparameters()
        try:                                                    # pragma: no cover - hydra is not in this image
            from hydra.utils import instantiate
        except Exception:                                       # noqa: BLE001
            from ..runtime.config import instantiate
        self.decoder = instantiate(decoder, input_channel=latent_dim, output_channel=self.channels, output_act=self.output_act)
        self.encoder = instantiate(encoder, input_channel=self.channels, output_channel=2 * latent_dim)
        if decoder_dist != "gaussian":
            raise NotImplementedError(f"decoder_dist={decoder_dist!r}: the reference's get_decode_dist knows 'gaussian' and 'bernoulli'; "
                                      "only the Gaussian decoder of the shipped configs is built here")
        self.decoder_dist = _GaussianDistribution()
        object.__setattr__(self, "_anchor", torch.zeros(1, requires_grad=True))

    def forward(self, z):
        """Generate images given latent code."""
        output = self.decoder_dist.sample(self.decoder(z))
        return output.reshape(output.shape[0], self.channels, self.height, self.width)

    def sample(self, N: int):
        z = torch.randn(N, self.hparams.latent_dim).to(self.device)
        return self.forward(z)

    def flat_nets(self):
        return [self.decoder, self.encoder]

    def configure_optimizers(self):
        from ..runtime.optim import FlatAdam
        hp = self.hparams
        opt = FlatAdam(self.flat_nets(), lr=hp.lr, betas=(hp.b1, hp.b2))
        scheduler = torch.optim.lr_scheduler.StepLR(opt, 1, gamma=0.99)
        return [opt], [scheduler]

    def _step(self, imgs, eps=None):
        if eps is None:
            eps = torch.randn(imgs.shape[0], self.hparams.latent_dim, device=imgs.device)      # where Normal.rsample() draws
        if self._anchor.device != imgs.device:
            object.__setattr__(self, "_anchor", torch.zeros(1, device=imgs.device, requires_grad=True))
        anchor = self._anchor if (torch.is_grad_enabled() and self.training) else self._anchor.detach()
        return _VAEStep.apply(imgs, eps, anchor, self)

    def vae(self, imgs):
        """(mu, log_sigma, z, recon_imgs) like the reference (vae.py:51-56); inference only -- training goes through training_step."""
        with torch.no_grad():
            n = imgs.shape[0]
            h, _ = self.encoder.forward_nhwc(K.nchw_to_nhwc(imgs.float()), record=False)
            hrows = h.reshape(n, -1)
            eps = torch.randn(n, self.hparams.latent_dim, device=imgs.device)
            z, _ = K.vae_latent_fwd(hrows, eps)
            out, _ = self.decoder.forward_nhwc(z.view(n, 1, 1, -1), record=False)
            mu, log_sigma = torch.chunk(hrows, 2, dim=1)
            return mu, log_sigma, z, K.nhwc_to_nchw(out)

    def training_step(self, batch, batch_idx, eps=None):
        imgs, labels = batch
        neg_elbo, kld, log_p_x_of_z, _, _ = self._step(imgs, eps)
        self.log("train_log/elbo", -neg_elbo.detach())
        self.log("train_log/kl_divergence", kld)
        self.log("train_log/log_p_x_of_z", log_p_x_of_z)
        return neg_elbo

    def validation_step(self, batch, batch_idx):
        imgs, labels = batch
        N = imgs.shape[0]
        with torch.no_grad():
            mu, log_sigma, z, recon_imgs = self.vae(imgs)
            chw = imgs[0].numel()
            mse, _ = K.eps_loss(K.nchw_to_nhwc(recon_imgs), imgs.float(), 1, want_grad=False)
            log_p_x_of_z = -0.5 * chw * mse - 0.5 * chw * math.log(2 * math.pi)
            fake_imgs = self.sample(N)
        self.log("val_log/log_p_x_of_z", log_p_x_of_z)
        return ValidationResult(real_image=imgs, fake_image=fake_imgs, recon_image=recon_imgs, label=labels, encode_latent=z)
