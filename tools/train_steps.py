#!/usr/bin/env python3
"""N cfg-2 training steps and nothing else (for rocprofv3 --kernel-trace --stats)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-generation-models_amd")]
import torch
from src.models.ddpm import DDPM
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
torch.manual_seed(0)
m = DDPM({"width": 32, "height": 32, "channels": 3, "transforms": {"normalize": True}}, hidden_dim=128, dim_mults=(1, 2, 4),
         lr=1e-4, b1=0.9, b2=0.999).to("cuda")
m.denoising_model.compute_mode = os.environ.get("MODE", "bf16"); m.train()
opt = m.configure_optimizers()
x = torch.rand(B, 3, 32, 32, device="cuda") * 2 - 1
for i in range(n):
    loss = m.training_step((x, None), i); loss.backward(); opt.step()
torch.cuda.synchronize()
print("loss", float(loss))
