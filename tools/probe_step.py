#!/usr/bin/env python3
"""One instrumented cfg-2 training step: per-launch time of every conv-family kernel (GPU box)."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-generation-models_amd")]
import torch
from src.models.ddpm import DDPM
from src.ops import functional as K
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
torch.manual_seed(0)
CFG3 = len(sys.argv) > 2 and sys.argv[2] == "cfg3"        # python tools/probe_step.py 32 cfg3
S = 64 if CFG3 else 32
m = DDPM({"width": S, "height": S, "channels": 3, "transforms": {"normalize": True}}, hidden_dim=64 if CFG3 else 128,
         dim_mults=(1, 2, 4, 8) if CFG3 else (1, 2, 4), lr=1e-4, b1=0.9, b2=0.999).to("cuda")
m.denoising_model.compute_mode = "bf16"; m.train()
opt = m.configure_optimizers()
x = torch.rand(B, 3, S, S, device="cuda") * 2 - 1
for i in range(3):
    loss = m.training_step((x, None), i); loss.backward(); opt.step()
torch.cuda.synchronize()
K.PROBE = []
loss = m.training_step((x, None), 0); loss.backward(); opt.step()
torch.cuda.synchronize()
agg = collections.OrderedDict()
for sym, fl, e0, e1, desc, *_ in K.PROBE:
    a = agg.setdefault((sym, desc), [0, 0.0, 0.0]); a[0] += 1; a[1] += e0.elapsed_time(e1) * 1e3; a[2] += fl
tot = sum(v[1] for v in agg.values())
print(f"total conv-family {tot/1e3:.2f} ms")
for (sym, desc), (n, us, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{us:8.1f}us x{n:2d} {fl/us/1e6:7.1f}TF  {sym:28s} {desc}")
