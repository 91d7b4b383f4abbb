"""CPU oracle of the VQ-VAE model -- TEST INFRASTRUCTURE ONLY.

Functional restatement (torch CPU ops over a plain state_dict) of
  * the encoder / decoder of `/root/reference/src/networks/vqvae.py:51-137` (incl. the weight-shared residual stack,
    :41-43, and the in-place ReLU that makes the skip connection carry relu(x), :16,:25),
  * `VQVAE.forward` and `VQVAE.training_step` of `/root/reference/src/models/vqvae.py:81-114`
    (straight-through decoder input :104, total = recon + vq + beta * commit with commit already weighted by beta, :39,:110).
Only tests/ (and benchmark CPU baselines) import it.  Pinned on tests/golden/vqvae_kats.npz, which
tools/gen_golden_vqvae.py produces by running the reference's own classes and training_step.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import vq_oracle as V


def _stack(x, sd, pre, n):
    w3, w1 = sd[pre + "stack.0.res_block.1.weight"], sd[pre + "stack.0.res_block.3.weight"]
    for _ in range(n):
        x = F.relu(x)                                    # nn.ReLU(True) acts on x itself before the add
        x = x + F.conv2d(F.relu(F.conv2d(x, w3, None, 1, 1)), w1)
    return F.relu(x)


def encoder(sd, x, pre="encoder.", n_res_layers=3):
    p = pre + "conv_stack."
    x = F.relu(F.conv2d(x, sd[p + "0.weight"], sd[p + "0.bias"], 2, 1))
    x = F.relu(F.conv2d(x, sd[p + "2.weight"], sd[p + "2.bias"], 2, 1))
    x = F.conv2d(x, sd[p + "4.weight"], sd[p + "4.bias"], 1, 1)
    return _stack(x, sd, p + "5.", n_res_layers)


def decoder(sd, z, pre="decoder.", n_res_layers=3):
    p = pre + "inverse_conv_stack."
    x = F.conv_transpose2d(z, sd[p + "0.weight"], sd[p + "0.bias"], 1, 1)
    x = _stack(x, sd, p + "1.", n_res_layers)
    x = F.relu(F.conv_transpose2d(x, sd[p + "2.weight"], sd[p + "2.bias"], 2, 1))
    return F.conv_transpose2d(x, sd[p + "4.weight"], sd[p + "4.bias"], 2, 1)


def forward(sd, imgs, beta, n_res_layers=3):
    z = encoder(sd, imgs, n_res_layers=n_res_layers)
    quant, _, _, _ = V.vq_forward(z, sd["vector_quntizer.embedding"], beta)
    return decoder(sd, quant, n_res_layers=n_res_layers)


def training_losses(sd, imgs, beta, n_res_layers=3):
    """(total, recon, vq, commit, idx) as VQVAE.training_step composes them (vqvae.py:94-110)."""
    z = encoder(sd, imgs, n_res_layers=n_res_layers)
    quant, vq_loss, commit_loss, idx = V.vq_forward(z, sd["vector_quntizer.embedding"], beta)
    dec_in = z + (quant - z).detach()
    recon = F.mse_loss(decoder(sd, dec_in, n_res_layers=n_res_layers), imgs)
    return recon + vq_loss + beta * commit_loss, recon, vq_loss, commit_loss, idx


def training_grads(sd, imgs, beta, n_res_layers=3):
    """Losses + autograd gradients for every distinct parameter (shared residual layers listed under index 0)."""
    leaf = {}
    for k, v in sd.items():
        if ".stack." in k and ".stack.0." not in k:
            continue
        leaf[k] = v.detach().clone().requires_grad_(True)
    full = dict(leaf)
    for k in sd:
        if k not in full:
            i = k.index(".stack.") + len(".stack.")
            full[k] = leaf[k[:i] + "0" + k[k.index(".", i):]]
    total, recon, vq_loss, commit_loss, idx = training_losses(full, imgs, beta, n_res_layers)
    total.backward()
    return (total.detach(), recon.detach(), vq_loss.detach(), commit_loss.detach(), idx), {k: v.grad for k, v in leaf.items()}
