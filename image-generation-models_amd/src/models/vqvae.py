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

    @torch.no_grad()
    def indices(self, z):
        """Codebook indices [N, H, W] (what a prior over the latents trains on)."""
        n, d, h, w = z.shape
        return _VQFunction.apply(z, self.embedding, self.commitment_weight)[3].view(n, h, w)
