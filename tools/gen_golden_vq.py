#!/usr/bin/env python3
"""Golden vectors for the VQ-VAE codebook step, produced by the REFERENCE VectorQuantizer itself.

Runs only in the build container (imports /root/reference/src/models/vqvae.py with import stubs for
hydra / pytorch_lightning / omegaconf, which the image lacks) and writes plain numpy arrays to
tests/golden/vq_kats.npz: inputs, codebook, the reference's quantised output, both losses, the
indices and the gradients autograd gives for (vq_loss + beta * commit_loss) and the straight-through
decoder input of vqvae.py:104.

    python tools/gen_golden_vq.py
"""
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

    pl = _stub("pytorch_lightning", LightningModule=_LM, LightningDataModule=object, Callback=object, Trainer=object,
               seed_everything=torch.manual_seed)
    pl.loggers = _stub("pytorch_lightning.loggers", Logger=object)
    pl.utilities = _stub("pytorch_lightning.utilities", rank_zero_only=lambda f: f)
    _stub("torchvision")
    _stub("hydra", utils=types.SimpleNamespace(instantiate=None))
    _stub("omegaconf", DictConfig=dict, OmegaConf=object)
    sys.path.insert(0, REF)
    from src.models import vqvae
    return vqvae


def main():
    ref = import_reference()
    out = {}
    for tag, (n, d, h, w, k, beta, scale) in {"small": (2, 8, 4, 4, 16, 0.25, 1.0), "cfg4": (4, 64, 16, 16, 512, 0.25, 0.05),
                                              "ragged": (3, 32, 5, 7, 40, 0.5, 0.3)}.items():
        torch.manual_seed(7)
        vq = ref.VectorQuantizer(k, d, beta)
        with torch.no_grad():
            vq.embedding.mul_(k * scale)                      # spread the codebook so that rows have distinct winners
        z = (torch.randn(n, d, h, w) * scale).requires_grad_(True)
        q, vq_loss, commit_loss = vq(z)
        rows = z.reshape(n, d, -1).permute(0, 2, 1).reshape(-1, d)
        idx = torch.argmin(torch.cdist(rows, vq.embedding), dim=1)
        (vq_loss + beta * commit_loss).backward()
        out.update({f"{tag}.z": z.detach().numpy(), f"{tag}.codebook": vq.embedding.detach().numpy(), f"{tag}.beta": np.float64(beta),
                    f"{tag}.quant": q.detach().numpy(), f"{tag}.vq_loss": np.float64(vq_loss.item()),
                    f"{tag}.commit_loss": np.float64(commit_loss.item()), f"{tag}.idx": idx.numpy().astype(np.int64),
                    f"{tag}.dz": z.grad.numpy(), f"{tag}.dcodebook": vq.embedding.grad.numpy()})
    np.savez_compressed(os.path.join(OUT, "vq_kats.npz"), **out)
    print("wrote", os.path.join(OUT, "vq_kats.npz"), {k: v.shape for k, v in out.items() if hasattr(v, "shape") and v.ndim})


if __name__ == "__main__":
    main()
