#!/usr/bin/env python3
"""Robustness check: three DDPM training steps at batch sizes the fast kernels do not tile evenly (5, 13, 210), fp32 vs bf16 losses.
   python tools/odd_batch.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-generation-models_amd")]
from src.models.ddpm import DDPM
for B in (5, 13, 210):
    res = {}
    for mode in ("fp32", "bf16"):
        torch.manual_seed(0)
        m = DDPM({"width": 32, "height": 32, "channels": 3, "transforms": {"normalize": True}}, hidden_dim=64, dim_mults=(1, 2, 4)).to("cuda")
        m.denoising_model.compute_mode = mode; m.train()
        opt = m.configure_optimizers()
        imgs = torch.rand(B, 3, 32, 32, device="cuda") * 2 - 1
        torch.manual_seed(1)
        losses = []
        for i in range(3):
            loss = m.training_step((imgs, None), i); loss.backward(); opt.step(); losses.append(float(loss.detach()))
        res[mode] = losses
    print(B, res)
