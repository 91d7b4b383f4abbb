#!/usr/bin/env python3
"""The named fused unit at level 0 (B=128, 32x32, 128 -> 128; STORAGE=bf16|fp32): plain pw conv, the GroupNorm pass, fused (coefficient tensor), fused (sums)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-generation-models_amd")]
import torch
from src.ops import functional as K
N, H, W, Cc = int(os.environ.get("B", 128)), 32, 32, 128
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(7)
gamma = torch.ones(Cc, device=dev); beta = torch.zeros(Cc, device=dev)
temb = torch.randn(N, Cc, device=dev, generator=g) * 0.1
w32 = torch.randn(9 * Cc * Cc, device=dev, generator=g) * 0.03
table, nent, tiles = K.pack_table([(0, 9, Cc, Cc)], dev)
wd, w, wdq, wq = (torch.zeros(w32.numel(), device=dev, dtype=torch.bfloat16) for _ in range(4))
K.pack_weights_bf16(table, nent, tiles, w32, wd, w, wdq, wq)
bias = torch.zeros(Cc, device=dev)
DT = torch.float32 if os.environ.get("STORAGE", "bf16") == "fp32" else torch.bfloat16
x = torch.randn(N, H, W, Cc, device=dev, generator=g).to(DT)
xs = x.float().view(N, H * W, Cc // 16, 16)
sums = K.gn_sums_encode(torch.stack([xs.sum((1, 3)), (xs * xs).sum((1, 3))], dim=-1))
stats, coef = K.gn_stats_coef(x, gamma, beta, temb=temb)
gn = (sums, gamma, beta, temb, 8, 1e-5)
fns = {"plain": lambda: K.conv3x3_bf16w(x, w, K=Cc, Nc=Cc, flip=False, bias=bias, out_dtype=DT, wq=wq),
       "gn": lambda: K.gn_mish_fwd(x, gamma, beta, temb=temb, out_dtype=DT),
       "fused_coef": lambda: K.conv3x3_gn_mish(x, coef, w, K=Cc, Nc=Cc, bias=bias, wq=wq),
       "fused_sums": lambda: K.conv3x3_gn_mish(x, None, w, K=Cc, Nc=Cc, bias=bias, gn=gn, wq=wq)}
res = {k: [] for k in fns}
for rnd in range(5):
    for k, fn in fns.items():
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        res[k].append(e0.elapsed_time(e1) / 20 * 1e3)
print(str(DT), {k: round(sorted(v)[2], 1) for k, v in res.items()})
