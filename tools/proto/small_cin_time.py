#!/usr/bin/env python3
"""Times the image -> features 3x3 conv (Conv2d(3, C, 3, padding=1)) forward and weight gradient at the cfg-2 / cfg-3 shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-generation-models_amd")]
import torch
from src.ops import functional as K
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (B, H, C) in ((128, 32, 128), (32, 64, 64)):
    x = torch.randn(B, H, H, 4, device="cuda")[..., :3]
    w = torch.randn(27 * C, device="cuda") * 0.1
    b = torch.randn(C, device="cuda")
    dW = torch.zeros(27 * C, device="cuda")
    for dt in (torch.bfloat16, torch.float32):
        if dt == torch.bfloat16 and not K.small_cin_bf16_supported(3, B, H, H, 3, C, 4):
            continue
        dy = torch.randn(B, H, H, C, device="cuda").to(dt)
        tf = timeit(lambda: K.conv_small_cin_fwd(x, w, b, C, 3, out_dtype=dt))
        tw = timeit(lambda: K.conv_small_cin_wgrad(x, dy, dW, 3))
        print(f"B{B} {H}x{H} C{C} {str(dt)[6:]:9s} fwd {tf:6.1f} us   wgrad {tw:6.1f} us", flush=True)
