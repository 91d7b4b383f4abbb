"""CPU oracle of the WGAN-GP path -- TEST INFRASTRUCTURE ONLY.

Functional restatement (torch CPU ops over plain state_dicts, autograd with create_graph for the penalty) of
  * the generator / critic of `/root/reference/src/networks/conv64.py:8-84` and `conv32.py:9-82` with
    `norm_type="layer"` = nn.GroupNorm(1, C) (`basic.py:33-37`; forced at `src/models/wgan_gp.py:30-31`),
  * both branches of `WGAN.training_step` (`wgan_gp.py:52-107`).
Only tests/ import it.  Pinned on tests/golden/wgan_kats.npz, written by tools/gen_golden_wgan.py from the reference's own
classes and training_step.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def generator(sd, z, pre=""):
    x = z.reshape(z.shape[0], -1, 1, 1)
    for i in range(4):
        x = F.conv_transpose2d(x, sd[f"{pre}main.{3 * i}.weight"], sd[f"{pre}main.{3 * i}.bias"], 1 if i == 0 else 2, 0 if i == 0 else 1)
        x = F.relu(F.group_norm(x, 1, sd[f"{pre}main.{3 * i + 1}.weight"], sd[f"{pre}main.{3 * i + 1}.bias"]))
    return torch.tanh(F.conv_transpose2d(x, sd[pre + "main.12.weight"], sd[pre + "main.12.bias"], 2, 1))


def critic(sd, x, pre=""):
    x = F.leaky_relu(F.conv2d(x, sd[pre + "main.0.weight"], sd[pre + "main.0.bias"], 2, 1), 0.2)
    for c, n in ((2, 3), (5, 6), (8, 9)):
        x = F.conv2d(x, sd[f"{pre}main.{c}.weight"], sd[f"{pre}main.{c}.bias"], 2, 1)
        x = F.leaky_relu(F.group_norm(x, 1, sd[f"{pre}main.{n}.weight"], sd[f"{pre}main.{n}.bias"]), 0.2)
    return F.conv2d(x, sd[pre + "main.11.weight"], sd[pre + "main.11.bias"], 1, 0).reshape(x.shape[0], -1)


def gradient_penalty(sd_d, inter):
    inter = inter.detach().requires_grad_(True)
    prob = critic(sd_d, inter)
    grads = torch.autograd.grad(prob, inter, torch.ones_like(prob), create_graph=True, retain_graph=True)[0]
    return torch.mean((torch.linalg.vector_norm(grads.reshape(inter.shape[0], -1), dim=1) - 1) ** 2), grads


def critic_step(sd_g, sd_d, imgs, z, lerp, gp_weight=10.0):
    """(d_loss, real_loss, fake_loss, penalty) with autograd attached to the entries of sd_d (wgan_gp.py:74-100)."""
    real_loss = -critic(sd_d, imgs).mean()
    fake = generator(sd_g, z).detach()
    fake_loss = critic(sd_d, fake).mean()
    pen, _ = gradient_penalty(sd_d, lerp * imgs + (1 - lerp) * fake)
    return real_loss + fake_loss + gp_weight * pen, real_loss, fake_loss, pen


def generator_step(sd_g, sd_d, z):
    return -critic(sd_d, generator(sd_g, z)).mean()
