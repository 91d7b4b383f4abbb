#!/usr/bin/env python3
"""Golden vectors for the VQ-VAE model, produced by the REFERENCE classes and the reference's own training_step.

Runs only in the build container: imports /root/reference/src/models/vqvae.py and src/networks/vqvae.py with import
stubs for hydra / pytorch_lightning / omegaconf (absent from the image) and writes plain arrays to
tests/golden/vqvae_kats.npz.

  tiny : latent 32, 32 codes, res_h_dim 32, 2x3x16x16 images -- full state_dict, images, VQVAE.forward output,
         the four losses of training_step, the code indices and every parameter gradient.
  cfg4 : configs/model/vqvae.yaml sizes (latent 64, 512 codes, defaults), 8x3x32x32 images, weights = the seeded
         default init (torch.manual_seed(1236), construction order decoder, encoder, codebook), codebook then scaled
         by 512 * 0.05 (smallest relative argmin gap 4.8e-4: no near-ties) -- images, per-parameter (sum, |sum|) of the weights, losses,
         indices, per-parameter gradient (sum, L2 norm), forward output.

  cfg4_64 : the same at BASELINE configs[3]'s own size, 4x3x64x64 images (seed 1240, smallest relative argmin gap 1.6e-4): same fields as cfg4.

    python tools/gen_golden_vqvae.py
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

        def log(self, *a, **k):
            pass

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
    from src.models import vqvae
    return vqvae


def build(ref, size, latent, codes, beta, enc_kw, dec_kw):
    dm = types.SimpleNamespace(width=size, height=size, channels=3, transforms=types.SimpleNamespace(normalize=True))
    m = ref.VQVAE(dm, encoder={"_target_": "src.networks.vqvae.Encoder", **enc_kw}, decoder={"_target_": "src.networks.vqvae.Decoder", **dec_kw},
                  latent_dim=latent, num_embeddings=codes, beta=beta)
    m.hparams = types.SimpleNamespace(beta=beta, lr=1e-3, b1=0.9, b2=0.999, latent_dim=latent)
    return m


def run(m, imgs):
    total = m.training_step((imgs, None), 0)
    total.backward()
    with torch.no_grad():
        z = m.encoder(imgs)
        rows = z.reshape(z.shape[0], z.shape[1], -1).permute(0, 2, 1).reshape(-1, z.shape[1])
        idx = torch.argmin(torch.cdist(rows, m.vector_quntizer.embedding), dim=1)
        q, vq_loss, commit_loss = m.vector_quntizer(z)
        recon = torch.nn.functional.mse_loss(m.decoder(q), imgs)
        out = m.forward(imgs)
        d2 = torch.cdist(rows, m.vector_quntizer.embedding) ** 2
        two = torch.topk(d2, 2, dim=1, largest=False).values
        print("   min relative argmin gap:", float(((two[:, 1] - two[:, 0]) / two[:, 1]).min()))
    return total.detach(), recon, vq_loss, commit_loss, idx, out


def main():
    ref = import_reference()
    out = {}
    # ---- tiny: everything stored
    torch.manual_seed(99)
    m = build(ref, 16, 32, 32, 0.25, {"res_h_dim": 32}, {"h_dim": 32, "res_h_dim": 32})
    with torch.no_grad():
        m.vector_quntizer.embedding.mul_(32 * 0.05)
    imgs = torch.rand(2, 3, 16, 16) * 2 - 1
    total, recon, vq_loss, commit_loss, idx, fwd = run(m, imgs)
    out["tiny.imgs"] = imgs.numpy()
    for k, v in m.state_dict().items():
        out["tiny.sd." + k] = v.detach().numpy()
    for k, p in m.named_parameters():
        out["tiny.grad." + k] = p.grad.numpy()
    out.update({"tiny.total": np.float64(total), "tiny.recon": np.float64(recon), "tiny.vq": np.float64(vq_loss),
                "tiny.commit": np.float64(commit_loss), "tiny.idx": idx.numpy().astype(np.int64), "tiny.forward": fwd.numpy()})
    # ---- cfg4: seeded default init, scalars only
    torch.manual_seed(1236)
    m = build(ref, 32, 64, 512, 0.25, {}, {})
    with torch.no_grad():
        m.vector_quntizer.embedding.mul_(512 * 0.05)
    imgs = torch.rand(8, 3, 32, 32) * 2 - 1
    total, recon, vq_loss, commit_loss, idx, fwd = run(m, imgs)
    out["cfg4.imgs"] = imgs.numpy()
    names = [k for k, _ in m.named_parameters()]
    out["cfg4.names"] = np.array(names)
    out["cfg4.wstats"] = np.array([[float(p.detach().double().sum()), float(p.detach().double().abs().sum())] for _, p in m.named_parameters()])
    out["cfg4.gstats"] = np.array([[float(p.grad.double().sum()), float(p.grad.double().norm())] for _, p in m.named_parameters()])
    out.update({"cfg4.total": np.float64(total), "cfg4.recon": np.float64(recon), "cfg4.vq": np.float64(vq_loss),
                "cfg4.commit": np.float64(commit_loss), "cfg4.idx": idx.numpy().astype(np.int64),
                "cfg4.forward": fwd.numpy().astype(np.float32)})
    # ---- cfg4_64: BASELINE configs[3] at its own size (CelebA 64x64 -> 16x16 latents, 256 rows per image)
    torch.manual_seed(1240)
    m = build(ref, 64, 64, 512, 0.25, {}, {})
    with torch.no_grad():
        m.vector_quntizer.embedding.mul_(512 * 0.05)
    imgs = torch.rand(4, 3, 64, 64) * 2 - 1
    total, recon, vq_loss, commit_loss, idx, fwd = run(m, imgs)
    out["cfg4_64.imgs"] = imgs.numpy()
    out["cfg4_64.wstats"] = np.array([[float(p.detach().double().sum()), float(p.detach().double().abs().sum())] for _, p in m.named_parameters()])
    out["cfg4_64.gstats"] = np.array([[float(p.grad.double().sum()), float(p.grad.double().norm())] for _, p in m.named_parameters()])
    out.update({"cfg4_64.total": np.float64(total), "cfg4_64.recon": np.float64(recon), "cfg4_64.vq": np.float64(vq_loss),
                "cfg4_64.commit": np.float64(commit_loss), "cfg4_64.idx": idx.numpy().astype(np.int64),
                "cfg4_64.forward": fwd.numpy().astype(np.float32)})
    np.savez_compressed(os.path.join(OUT, "vqvae_kats.npz"), **out)
    print("wrote vqvae_kats.npz:", {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items() if not k.startswith("tiny.sd.") and not k.startswith("tiny.grad.")})
    print("names:", names)


if __name__ == "__main__":
    main()
