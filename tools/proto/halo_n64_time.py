#!/usr/bin/env python3
"""3x3 conv on cfg 3's 64 x 64 level (64 / 128 -> 64 channels, B = 32), bf16 storage: the halo kernel with all waves along M (N64)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-generation-models_amd")]
import torch
from src.ops import functional as K
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B, H = 32, 64
for Ci, Co in ((64, 64), (128, 64), (64, 128)):
    x = torch.randn(B, H, H, Ci, device="cuda").bfloat16()
    w = (torch.randn(9 * Co * Ci, device="cuda") * 0.05).bfloat16()
    y = torch.empty(B, H, H, Co, device="cuda", dtype=torch.bfloat16)
    yf = torch.zeros(B, H, H, Co, device="cuda")
    fl = 2.0 * B * H * H * Ci * Co * 9
    t = timeit(lambda: K.conv3x3_bf16w(x, w, K=Ci, Nc=Co, flip=False, out=y))
    t2 = timeit(lambda: K.conv3x3_bf16w(x, w, K=Ci, Nc=Co, flip=True, out=yf, accumulate=True))
    print(f"{Ci}->{Co} @64x64 B{B}: bf16 out {t:6.1f} us {fl/t/1e6:6.0f} TF | fp32 accumulate {t2:6.1f} us {fl/t2/1e6:6.0f} TF", flush=True)
