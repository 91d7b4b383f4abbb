"""CPU oracle of the VAE path (BASELINE cfg 1) -- TEST INFRASTRUCTURE ONLY.

Functional restatement (torch CPU ops over plain state_dicts) of
  * `ConvEncoder` / `ConvDecoder` of `/root/reference/src/networks/basic.py:147-204` with `norm_type="batch"` (training-mode
    nn.BatchNorm2d: batch statistics, biased variance in the normalisation) or evaluation mode (running statistics),
  * `VAE.vae` / `VAE.training_step` of `/root/reference/src/models/vae.py:46-72`, the unit-variance Gaussian decoder of
    `src/utils/distributions.py:12-24` and `normal_kld` of `src/utils/losses.py:30-32`.
Only tests/ import it.  Pinned on tests/golden/vae_kats.npz (tools/gen_golden_vae.py: the reference's own training_step).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def _bn(x, sd, pre, training):
    if training:
        return F.batch_norm(x, None, None, sd[pre + "weight"], sd[pre + "bias"], True, 0.1, 1e-5)
    return F.batch_norm(x, sd[pre + "running_mean"], sd[pre + "running_var"], sd[pre + "weight"], sd[pre + "bias"], False, 0.1, 1e-5)


def encoder(sd, x, pre="encoder.", training=True):
    p = pre + "network."
    x = F.leaky_relu(F.conv2d(x, sd[p + "0.weight"], sd[p + "0.bias"], 2, 1), 0.2)
    x = F.leaky_relu(_bn(F.conv2d(x, sd[p + "2.weight"], sd[p + "2.bias"], 2, 1), sd, p + "3.", training), 0.2)
    x = F.leaky_relu(_bn(F.conv2d(x, sd[p + "5.weight"], sd[p + "5.bias"], 2, 1), sd, p + "6.", training), 0.2)
    return F.conv2d(x, sd[p + "8.weight"], sd[p + "8.bias"], 1, 0).reshape(x.shape[0], -1)


def decoder(sd, z, pre="decoder.", training=True):
    p = pre + "network."
    x = z.reshape(z.shape[0], -1, 1, 1)
    for i, (s, pad) in enumerate(((1, 0), (2, 1), (2, 1))):
        x = F.conv_transpose2d(x, sd[f"{p}{3 * i}.weight"], sd[f"{p}{3 * i}.bias"], s, pad)
        x = F.relu(_bn(x, sd, f"{p}{3 * i + 1}.", training))
    return torch.tanh(F.conv_transpose2d(x, sd[p + "9.weight"], sd[p + "9.bias"], 2, 1))


def training_losses(sd, imgs, eps, beta=1.0, recon_weight=1.0):
    """(-elbo, kld, log_p_x_of_z, z, recon) as VAE.training_step composes them."""
    h = encoder(sd, imgs)
    mu, log_sigma = torch.chunk(h, 2, dim=1)
    z = mu + torch.exp(log_sigma) * eps
    recon = decoder(sd, z)
    kld = (-0.5 * torch.sum(1 + 2 * log_sigma - mu ** 2 - torch.exp(2 * log_sigma), dim=-1)).mean(dim=0)
    log_p = (-0.5 * (imgs - recon) ** 2 - 0.5 * math.log(2 * math.pi)).sum(dim=[1, 2, 3]).mean(dim=0)
    return -(-beta * kld + recon_weight * log_p), kld, log_p, z, recon
