#!/usr/bin/env python3
"""Stress: many training steps (eager or one hipGraph), every loss checked for finiteness; on the first non-finite loss reports the
step and which gradient / parameter tensors are non-finite.   python tools/nan_hunt.py cfg2|cfg3 B steps [graph]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-generation-models_amd")]
import torch
from src.models.ddpm import DDPM

cfg, B, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
graph = len(sys.argv) > 4 and sys.argv[4] == "graph"
S, hd, mults = (64, 64, (1, 2, 4, 8)) if cfg == "cfg3" else (32, 128, (1, 2, 4))
torch.manual_seed(int(os.environ.get("SEED", "0")))
m = DDPM({"width": S, "height": S, "channels": 3, "transforms": {"normalize": True}}, hidden_dim=hd, dim_mults=mults, lr=1e-4, b1=0.9, b2=0.999).to("cuda")
m.denoising_model.compute_mode = "bf16"; m.train()
opt = m.configure_optimizers()
x = torch.rand(B, 3, S, S, device="cuda") * 2 - 1
if graph:
    opt.device_state = True
for i in range(3):
    loss = m.training_step((x, None), i); loss.backward(); opt.step()
if graph:
    from src.runtime.graphed import GraphedTrainStep
    m.log = lambda *a, **k: None
    gs = GraphedTrainStep(m, opt, (x, None), warmup=0)
    step = lambda i: gs((x, None))
else:
    def step(i):
        loss = m.training_step((x, None), i); loss.backward(); opt.step()
        return loss
losses = []
for i in range(n):
    losses.append(step(i).detach().clone())
    if i % 50 == 49 or i == n - 1:
        v = torch.stack(losses).float().cpu()
        bad = (~torch.isfinite(v)).nonzero().flatten().tolist()
        if bad:
            print(f"{cfg} B={B} {'graph' if graph else 'eager'}: NON-FINITE loss at step {i - len(losses) + 1 + bad[0]} (window of {len(losses)})")
            u = m.denoising_model
            print("   non-finite flat_params:", int((~torch.isfinite(u.flat_params)).sum()), " flat_grads:", int((~torch.isfinite(u.flat_grads)).sum()))
            sys.exit(1)
        losses = []
print(f"{cfg} B={B} {'graph' if graph else 'eager'}: {n} steps, all losses finite, last {float(v[-1]):.4f}")
