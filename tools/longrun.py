#!/usr/bin/env python3
"""Training-dynamics sanity: 400 DDPM steps (hidden 64, mults 1-2-4, B=128) on a small learnable synthetic distribution in fp32 and bf16
mode -- the two loss curves should fall together and the weights stay finite.   python tools/longrun.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-generation-models_amd")]
from src.models.ddpm import DDPM
for mode in ("fp32", "bf16"):
    torch.manual_seed(0)
    m = DDPM({"width": 32, "height": 32, "channels": 3, "transforms": {"normalize": True}}, hidden_dim=64, dim_mults=(1, 2, 4), lr=2e-4, b1=0.9, b2=0.999).to("cuda")
    m.denoising_model.compute_mode = mode; m.train()
    opt = m.configure_optimizers()
    g = torch.Generator(device="cuda").manual_seed(5)
    # a learnable synthetic distribution: smooth blobs, fixed set of 512 images
    base = torch.nn.functional.interpolate(torch.randn(512, 3, 4, 4, device="cuda", generator=g), size=32, mode="bilinear").clamp(-1, 1)
    torch.manual_seed(1)
    hist = []
    for i in range(400):
        idx = torch.randint(0, 512, (128,), device="cuda")
        loss = m.training_step((base[idx], None), i); loss.backward(); opt.step()
        if i % 50 == 0 or i == 399: hist.append(round(float(loss.detach()), 4))
    print(mode, hist, "finite params:", bool(torch.isfinite(m.denoising_model.flat_params).all()))
