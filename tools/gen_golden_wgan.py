#!/usr/bin/env python3
"""Golden vectors for the WGAN-GP path, produced by the REFERENCE WGAN.training_step (both branches).

Runs only in the build container: imports /root/reference/src/models/wgan_gp.py and src/networks/conv64.py / conv32.py with
import stubs for hydra / pytorch_lightning / torchvision (absent from the image) and writes plain arrays to
tests/golden/wgan_kats.npz.  For each case (c64: conv64 nets, ndf = ngf = 8, latent 16, 4 x 3 x 64 x 64; c32: conv32 nets,
ndf = ngf = 8, latent 12, 3 x 3 x 32 x 32): initial state_dicts, images, the seed set right before each training_step (the
step draws z and the interpolation weights from torch's global generator), the drawn z / weights, logged scalars,
the critic's gradients and post-step weights of a critic step, the generator's gradients of a generator step.

    python tools/gen_golden_wgan.py
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def import_reference():
    class _LM(torch.nn.Module):
        def save_hyperparameters(self, *a, **k):
            pass

        def log(self, k, v, *a, **kw):
            self.logged[k] = float(v)

        def optimizers(self):
            return self._opts

        def manual_backward(self, loss):
            loss.backward()

        @property
        def device(self):
            return torch.device("cpu")

    def instantiate(cfg, **kw):
        mod, _, name = cfg["_target_"].rpartition(".")
        params = {k: v for k, v in cfg.items() if k != "_target_"}
        params.update(kw)
        return getattr(importlib.import_module(mod), name)(**params)

    pl = _stub("pytorch_lightning", LightningModule=_LM, LightningDataModule=object, Callback=object, Trainer=object,
               seed_everything=torch.manual_seed)
    pl.loggers = _stub("pytorch_lightning.loggers", Logger=object)
    pl.utilities = _stub("pytorch_lightning.utilities", rank_zero_only=lambda f: f)
    _stub("torchvision")
    _stub("hydra", utils=types.SimpleNamespace(instantiate=instantiate))
    _stub("omegaconf", DictConfig=dict, OmegaConf=object)
    sys.path.insert(0, REF)
    from src.models import wgan_gp
    return wgan_gp


def main():
    ref = import_reference()
    out = {}
    for tag, (net, size, latent, n) in {"c64": ("conv64", 64, 16, 4), "c32": ("conv32", 32, 12, 3)}.items():
        torch.manual_seed(21)
        dm = types.SimpleNamespace(width=size, height=size, channels=3, transforms=types.SimpleNamespace(normalize=True))
        m = ref.WGAN(dm, netG={"_target_": f"src.networks.{net}.Decoder", "ngf": 8}, netD={"_target_": f"src.networks.{net}.Encoder", "ndf": 8},
                     latent_dim=latent)
        m.hparams = types.SimpleNamespace(latent_dim=latent, n_critic=5, lrG=1e-4, lrD=1e-4, b1=0.0, b2=0.9, gp_weight=10)
        m.logged = {}
        with torch.no_grad():                                  # default init gives a nearly constant critic: perturb every tensor
            for p in m.parameters():
                p.add_(torch.randn_like(p) * 0.05)
        m._opts = m.configure_optimizers()
        imgs = torch.rand(n, 3, size, size) * 2 - 1
        out[f"{tag}.imgs"] = imgs.numpy()
        for k, v in m.state_dict().items():
            out[f"{tag}.sd0.{k}"] = v.detach().numpy().copy()
        # critic step (batch_idx 0)
        torch.manual_seed(77)
        out[f"{tag}.z_c"] = torch.randn(n, latent).numpy()
        out[f"{tag}.lerp"] = torch.zeros(n, 1, 1, 1).uniform_().numpy()
        torch.manual_seed(77)
        m.training_step((imgs, None), 0)
        for k in ("train_loss/d_loss", "train_log/real_logit", "train_log/fake_logit", "train_log/gradient_panelty"):
            out[f"{tag}.log.{k}"] = np.float64(m.logged[k])
        for k, p in m.discriminator.named_parameters():
            out[f"{tag}.dgrad.{k}"] = p.grad.numpy().copy()
            out[f"{tag}.dpost.{k}"] = p.detach().numpy().copy()
        # generator step (batch_idx == n_critic) from the post-critic-step weights
        torch.manual_seed(78)
        out[f"{tag}.z_g"] = torch.randn(n, latent).numpy()
        torch.manual_seed(78)
        m.training_step((imgs, None), 5)
        out[f"{tag}.log.train_loss/g_loss"] = np.float64(m.logged["train_loss/g_loss"])
        for k, p in m.generator.named_parameters():
            out[f"{tag}.ggrad.{k}"] = p.grad.numpy().copy()
        print(tag, {k: v for k, v in m.logged.items()})
    np.savez_compressed(os.path.join(OUT, "wgan_kats.npz"), **out)
    print("wrote wgan_kats.npz", os.path.getsize(os.path.join(OUT, "wgan_kats.npz")))


if __name__ == "__main__":
    main()
