"""`VectorQuantizer` with the reference's surface (src/models/vqvae.py:13-43) over the HIP codebook kernels.

Same constructor (num_embeddings, latent_dim, commitment_weight), same parameter name (`embedding`, so the
reference's state_dict keys load), same init (uniform in +-1/num_embeddings), same forward contract:
z [N, latent_dim, H, W] -> (quant_z [N, latent_dim, H, W], vq_loss, commit_loss) with autograd flowing to z
through commit_loss and to the codebook through vq_loss; quant_z carries the gradient torch's gather would give
the codebook (the reference detaches it in VQVAE.training_step, vqvae.py:104, but the class itself does not).

The compute is mi_vq_nearest_fwd / mi_vq_bwd (include/mi_ddpm.h); there is no torch fallback: a CPU tensor or a
missing library raises.
"""
import torch
from torch import nn

from ..ops import functional as K
from .base import BaseModel, ValidationResult


class _VQFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, codebook, commitment_weight):
        n, d, h, w = z.shape
        rows = K.nchw_to_nhwc(z.float()).reshape(n * h * w, -1)              # (N*H*W, ld >= D) view, vqvae.py:28-32
        rows = rows[:, :d]
        idx, zq, ssum = K.vq_nearest(rows, codebook)
        mse = ssum / float(rows.shape[0] * d)
        ctx.save_for_backward(rows, codebook, idx)
        ctx.shape = (n, d, h, w)
        ctx.cw = float(commitment_weight)
        ctx.mark_non_differentiable(idx)
        ctx.set_materialize_grads(False)
        return K.nhwc_to_nchw(zq.view(n, h, w, d)), mse, ctx.cw * mse, idx

    @staticmethod
    def backward(ctx, dquant, g_vq, g_commit, _):
        rows, codebook, idx = ctx.saved_tensors
        n, d, h, w = ctx.shape
        dz = dcb = None
        if ctx.needs_input_grad[0]:
            dz = torch.empty((rows.shape[0], d), device=rows.device, dtype=torch.float32)
        if ctx.needs_input_grad[1]:
            dcb = torch.zeros_like(codebook)
        if (dz is not None or dcb is not None) and (g_vq is not None or g_commit is not None):
            zero = torch.zeros((), device=rows.device)
            g_dev = torch.stack([(g_vq if g_vq is not None else zero).float().reshape(()),
                                 (g_commit if g_commit is not None else zero).float().reshape(())])     # stays on the device
            K.vq_backward(rows, codebook, idx, 1.0, ctx.cw, dz=dz, dcodebook=dcb, g_dev=g_dev)
        elif dz is not None:
            dz.zero_()
        if dcb is not None and dquant is not None:           # quant_z differentiated directly: the gather's own gradient
            K.vq_scatter_rows(K.nchw_to_nhwc(dquant.float()).reshape(n * h * w, -1)[:, :d], idx, dcb)
        return (K.nhwc_to_nchw(dz.view(n, h, w, d)) if dz is not None else None), dcb, None


class VectorQuantizer(nn.Module):
    def __init__(self, num_embeddings, latent_dim, commitment_weight) -> None:
        super().__init__()
        self.embedding = nn.Parameter(torch.zeros(num_embeddings, latent_dim).uniform_(-1 / num_embeddings, 1 / num_embeddings))
        self.latent_dim = latent_dim
        self.commitment_weight = commitment_weight

    def forward(self, z):
        quant_z, vq_loss, commit_loss, _ = _VQFunction.apply(z, self.embedding, self.commitment_weight)
        return quant_z, vq_loss, commit_loss

    # flat-buffer contract of the fused Adam (src/runtime/optim.py)
    @property
    def flat_params(self):
        return self.embedding.data.view(-1)

    @property
    def flat_grads(self):
        if self.embedding.grad is None or self.embedding.grad.device != self.embedding.device:
            self.embedding.grad = torch.zeros_like(self.embedding.data)
        return self.embedding.grad.view(-1)

    def mark_params_dirty(self):
        pass

    @torch.no_grad()
    def indices(self, z):
        """Codebook indices [N, H, W] (what a prior over the latents trains on)."""
        n, d, h, w = z.shape
        return _VQFunction.apply(z, self.embedding, self.commitment_weight)[3].view(n, h, w)


class _VQVAEStep(torch.autograd.Function):
    """encoder -> codebook -> decoder -> the three losses of VQVAE.training_step (vqvae.py:94-110) as ONE autograd node:
    everything stays NHWC on the HIP kernels, the straight-through estimator (:104) is the decoder's input gradient
    handed to the encoder together with the commitment term."""

    @staticmethod
    def forward(ctx, imgs, anchor, model):
        enc, dec, vq = model.encoder, model.decoder, model.vector_quntizer
        beta = float(model.hparams.beta)
        record = ctx.needs_input_grad[1]
        imgs = imgs.float().contiguous()
        z, tape_e = enc.forward_nhwc(K.nchw_to_nhwc(imgs), record=record)
        n, h, w, d = z.shape
        rows = z.view(n * h * w, d)
        idx, zq, ssum = K.vq_nearest(rows, vq.embedding.data)
        out, tape_d = dec.forward_nhwc(zq.view(n, h, w, d), record=record)
        recon, dout = K.eps_loss(out, imgs, 1, want_grad=record)
        vq_loss = ssum / float(rows.numel())
        commit_loss = vq.commitment_weight * vq_loss
        ctx.model, ctx.tapes, ctx.saved = model, (tape_e, tape_d), (rows, idx, dout)
        ctx.mark_non_differentiable(recon, vq_loss, commit_loss)
        return recon + vq_loss + beta * commit_loss, recon, vq_loss, commit_loss

    @staticmethod
    def backward(ctx, dloss, *_):
        model = ctx.model
        enc, dec, vq = model.encoder, model.decoder, model.vector_quntizer
        tape_e, tape_d = ctx.tapes
        rows, idx, dout = ctx.saved
        ctx.tapes = ctx.saved = None
        K.scale_by_device_scalar(dout, dloss)
        dz = dec.backward_nhwc(tape_d, dout, need_dx=True)                 # d total / d decoder_z == d total / d encoder_z (:104)
        gcb = vq.flat_grads.view_as(vq.embedding)
        if not getattr(vq, "accumulate_grads", False):
            gcb.zero_()
        beta = float(model.hparams.beta)
        n, h, w, d = dz.shape
        K.vq_backward(rows, vq.embedding.data, idx, 1.0, beta * vq.commitment_weight, dz=dz.view(n * h * w, d), accumulate=True,
                      dcodebook=gcb, g_dev=torch.stack([dloss.reshape(()), dloss.reshape(())]).float())
        enc.backward_nhwc(tape_e, dz, need_dx=False)
        return None, None, None


class VQVAE(BaseModel):
    """Reference `VQVAE` (vqvae.py:45-141): same constructor, attribute names (`vector_quntizer` sic), hooks and logged keys.
    `encoder` / `decoder` are config nodes with `_target_` (configs/networks/vqvae.yaml), instantiated with the same
    keyword arguments the reference passes."""

    def __init__(self, datamodule, encoder=None, decoder=None, latent_dim=100, lr: float = 0.0002, b1: float = 0.5,
                 b2: float = 0.999, num_embeddings: int = 512, beta: float = 0.25, optim="adam", **kwargs):
        super().__init__(datamodule)
        self.save_hyperparameters()
        try:                                                    # pragma: no cover - hydra is not in this image
            from hydra.utils import instantiate
        except Exception:                                       # noqa: BLE001
            from ..runtime.config import instantiate
        self.decoder = instantiate(decoder, input_channel=latent_dim, output_channel=self.channels)
        self.encoder = instantiate(encoder, input_channel=self.channels, output_channel=latent_dim)
        self.vector_quntizer = VectorQuantizer(num_embeddings, latent_dim, beta)
        self.latent_w = self.width // 4
        self.latent_h = self.height // 4
        self.latent_size = self.latent_h * self.latent_w
        object.__setattr__(self, "_anchor", torch.zeros(1, requires_grad=True))

    def forward(self, imgs):
        z = self.encoder(imgs)
        quant_z, _, _ = self.vector_quntizer(z)
        output = self.decoder(quant_z)
        return output.reshape(output.shape[0], self.channels, self.height, self.width)

    def training_step(self, batch, batch_idx):
        imgs, _ = batch
        if self._anchor.device != imgs.device:
            object.__setattr__(self, "_anchor", torch.zeros(1, device=imgs.device, requires_grad=True))
        anchor = self._anchor if torch.is_grad_enabled() else self._anchor.detach()
        total_loss, recon_loss, vq_loss, commit_loss = _VQVAEStep.apply(imgs, anchor, self)
        self.log("train_loss/vq_loss", vq_loss)                # device scalars: no per-step host sync
        self.log("train_loss/recon_loss", recon_loss)
        self.log("train_loss/commit_loss", commit_loss)
        return total_loss

    def flat_nets(self):
        """The three flat parameter buffers (optimizer and data-parallel reducer walk this list)."""
        return [self.encoder, self.decoder, self.vector_quntizer]

    def configure_optimizers(self):
        from ..runtime.optim import FlatAdam
        hp = self.hparams
        return FlatAdam(self.flat_nets(), lr=hp.lr, betas=(hp.b1, hp.b2))

    def validation_step(self, batch, batch_idx):
        imgs, labels = batch
        with torch.no_grad():
            recon_imgs = self.forward(imgs)
            out_nhwc = K.nchw_to_nhwc(recon_imgs)
            self.log("val/recon_loss", K.eps_loss(out_nhwc, imgs.float(), 1, want_grad=False)[0])
        return ValidationResult(real_image=imgs, recon_image=recon_imgs)
