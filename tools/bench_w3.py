#!/usr/bin/env python3
"""wgrad3x3 on the level-0 / level-1 / level-2 shapes, bf16 operands (for ablation builds: MI_DDPM_LIB=...)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-generation-models_amd")]
import torch
from src.ops import functional as K
B = 128
for H, C in [(32, 128), (16, 256), (8, 512)]:
    x = torch.randn(B, H, H, C, device="cuda").bfloat16(); dy = torch.randn(B, H, H, C, device="cuda").bfloat16()
    dW = torch.zeros(9 * C * C, device="cuda")
    run = lambda: K.conv_wgrad(x, dy, dW, kh=3, kw=3, stride=1, pad=1, gather_i=True, Ci=C, Cj=C, grid_g=(H, H), grid_d=(H, H), mode=1)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    print(f"{H}x{H} C{C}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us (kernel + reduce)", flush=True)
