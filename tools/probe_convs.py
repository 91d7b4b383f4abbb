#!/usr/bin/env python3
"""Per-launch conv-family timings (HIP events) of one VQ-VAE step or one WGAN-GP critic + generator step, slowest first.
   python tools/probe_convs.py vqvae|wgan [fp32|bf16]"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
K = importlib.import_module("image-generation-models_amd.src.ops.functional")


def main():
    which, mode = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "fp32")
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    if which == "vqvae":
        M = importlib.import_module("image-generation-models_amd.src.models.vqvae")
        dm = {"width": 32, "height": 32, "channels": 3, "transforms": {"normalize": True}}
        m = M.VQVAE(dm, encoder={"_target_": "src.networks.vqvae.Encoder"}, decoder={"_target_": "src.networks.vqvae.Decoder"}, latent_dim=64).to(dev)
        nets, imgs = [m.encoder, m.decoder], torch.rand(128, 3, 32, 32, device=dev) * 2 - 1
        steps = [lambda: m.training_step((imgs, None), 0).backward()]
    else:
        M = importlib.import_module("image-generation-models_amd.src.models.wgan_gp")
        dm = {"width": 64, "height": 64, "channels": 3, "transforms": {"normalize": True}}
        m = M.WGAN(dm, netG={"_target_": "src.networks.conv64.Decoder", "ngf": 64}, netD={"_target_": "src.networks.conv64.Encoder", "ndf": 64}).to(dev)
        nets, imgs = [m.generator, m.discriminator], torch.rand(64, 3, 64, 64, device=dev) * 2 - 1
        steps = [lambda: m.training_step((imgs, None), 0), lambda: m.training_step((imgs, None), 5)]
    for n in nets:
        n.compute_mode = mode
    m.train()
    for s in steps:
        for _ in range(3):
            s()
    for i, s in enumerate(steps):
        K.PROBE = []
        s()
        torch.cuda.synchronize()
        rows = sorted(((e0.elapsed_time(e1) * 1e3, fl, sym, desc) for sym, fl, e0, e1, desc, *_ in K.PROBE), reverse=True)
        K.PROBE = None
        print(f"== step kind {i}: {len(rows)} conv launches, {sum(r[0] for r in rows):.0f} us")
        for us, fl, sym, desc in rows[:14]:
            print(f"{us:8.1f} us {fl / us / 1e6:7.1f} TF  {sym:34s} {desc}")


if __name__ == "__main__":
    main()
