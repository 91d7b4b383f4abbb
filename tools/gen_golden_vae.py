#!/usr/bin/env python3
"""Golden vectors for the VAE path (BASELINE cfg 1), produced by the REFERENCE VAE.training_step.

Runs only in the build container: imports /root/reference/src/models/vae.py and src/networks/basic.py with import stubs for
hydra / pytorch_lightning / omegaconf / torchvision and writes plain arrays to tests/golden/vae_kats.npz:
  tiny : ndf = ngf = 8, latent 16, 6 x 1 x 28 x 28 images, every tensor perturbed away from the default init -- initial
         state_dict, images, the reparameterisation noise recovered from (z - mu) / sigma, the three logged scalars, every
         parameter gradient, the batch-norm buffers after the step, an evaluation-mode decode of a fixed z (running statistics);
  cfg1 : configs/model/vae.yaml + configs/networks/conv_mnist.yaml sizes (latent 128, ndf = ngf = 32, batch norm), 16 images,
         weights = the seeded default init (torch.manual_seed(32)) -- images, noise, scalars, per-parameter weight / gradient
         statistics, buffers, the evaluation-mode decode.

    python tools/gen_golden_vae.py
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
    hy = _stub("hydra", utils=types.SimpleNamespace(instantiate=instantiate))
    sys.modules["hydra.utils"] = types.ModuleType("hydra.utils"); sys.modules["hydra.utils"].instantiate = instantiate
    _stub("omegaconf", DictConfig=dict, OmegaConf=object)
    sys.path.insert(0, REF)
    from src.models import vae
    return vae


def run_case(ref, tag, ndf, latent, n, seed, out, full):
    torch.manual_seed(seed)
    dm = types.SimpleNamespace(width=28, height=28, channels=1, transforms=types.SimpleNamespace(normalize=True))
    m = ref.VAE(dm, encoder={"_target_": "src.networks.basic.ConvEncoder", "ndf": ndf, "norm_type": "batch"},
                decoder={"_target_": "src.networks.basic.ConvDecoder", "ngf": ndf, "norm_type": "batch"}, latent_dim=latent, decoder_dist="gaussian")
    m.hparams = types.SimpleNamespace(latent_dim=latent, beta=1.0, recon_weight=1.0, lr=1e-4, b1=0.9, b2=0.999)
    m.logged = {}
    if full:                                                # tiny case: every tensor away from the default init, stored
        with torch.no_grad():
            for p in m.parameters():
                p.add_(torch.randn_like(p) * 0.03)
        for k, v in m.state_dict().items():
            out[f"{tag}.sd0.{k}"] = v.detach().numpy().copy()
    m.train()
    imgs = torch.rand(n, 1, 28, 28) * 2 - 1
    out[f"{tag}.imgs"] = imgs.numpy()
    # the noise the step will draw: run vae() once with the same seed (restoring the batch-norm buffers afterwards)
    keep = {k: v.clone() for k, v in m.state_dict().items()}
    torch.manual_seed(55)
    with torch.no_grad():
        mu, ls, z, _ = m.vae(imgs)
    out[f"{tag}.eps"] = ((z - mu) / torch.exp(ls)).numpy()
    m.load_state_dict(keep)
    torch.manual_seed(55)
    loss = m.training_step((imgs, None), 0)
    loss.backward()
    out[f"{tag}.loss"] = np.float64(loss.item())
    for k in ("train_log/elbo", "train_log/kl_divergence", "train_log/log_p_x_of_z"):
        out[f"{tag}.log.{k}"] = np.float64(m.logged[k])
    names = [k for k, _ in m.named_parameters()]
    out[f"{tag}.names"] = np.array(names)
    out[f"{tag}.wstats"] = np.array([[float(p.detach().double().sum()), float(p.detach().double().abs().sum())] for _, p in m.named_parameters()])
    out[f"{tag}.gstats"] = np.array([[float(p.grad.double().sum()), float(p.grad.double().norm())] for _, p in m.named_parameters()])
    if full:
        for k, p in m.named_parameters():
            out[f"{tag}.grad.{k}"] = p.grad.numpy().copy()
    for k, v in m.state_dict().items():
        if "running_" in k or "num_batches" in k:
            out[f"{tag}.buf1.{k}"] = v.detach().numpy().copy()
    m.eval()
    zfix = torch.randn(5, latent)
    out[f"{tag}.zfix"] = zfix.numpy()
    with torch.no_grad():
        out[f"{tag}.decode_eval"] = m(zfix).numpy()
    print(tag, {k: v for k, v in m.logged.items()}, "loss", float(loss))


def main():
    ref = import_reference()
    out = {}
    run_case(ref, "tiny", 8, 16, 6, 31, out, True)           # everything stored
    run_case(ref, "cfg1", 32, 128, 16, 32, out, False)       # configs/model/vae.yaml + conv_mnist.yaml sizes, seeded default init: scalars
    np.savez_compressed(os.path.join(OUT, "vae_kats.npz"), **out)
    print("wrote vae_kats.npz", os.path.getsize(os.path.join(OUT, "vae_kats.npz")))


if __name__ == "__main__":
    main()
