#!/usr/bin/env python3
"""What the GroupNorm sums cost conv_pw's epilogue (level 0, B=128, 128 -> 128, bf16 in): plain vs + sums, bf16 and fp32 output."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-generation-models_amd")]
import torch
from src.ops import functional as K
N, H, Cc = int(os.environ.get("B", 128)), int(os.environ.get("H", 32)), int(os.environ.get("C", 128))
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(7)
w32 = torch.randn(9 * Cc * Cc, device=dev, generator=g) * 0.03
table, nent, tiles = K.pack_table([(0, 9, Cc, Cc)], dev)
wd, w, wdq, wq = (torch.zeros(w32.numel(), device=dev, dtype=torch.bfloat16) for _ in range(4))
K.pack_weights_bf16(table, nent, tiles, w32, wd, w, wdq, wq)
bias = torch.zeros(Cc, device=dev)
x = torch.randn(N, H, H, Cc, device=dev, generator=g).bfloat16()
scratch = K.gn_sums_buffer(N, Cc, dev)
fns = {}
for name, dt in (("bf16", torch.bfloat16), ("fp32", torch.float32)):
    fns[name] = (lambda dt=dt: K.conv3x3_bf16w(x, w, K=Cc, Nc=Cc, flip=False, bias=bias, out_dtype=dt, wq=wq))
    fns[name + "+sums"] = (lambda dt=dt: K.conv3x3_bf16w(x, w, K=Cc, Nc=Cc, flip=False, bias=bias, out_dtype=dt, gn_sums=scratch, wq=wq))
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
print(os.environ.get("MI_DDPM_LIB", "default")[-24:], {k: round(sorted(v)[2], 1) for k, v in res.items()})
