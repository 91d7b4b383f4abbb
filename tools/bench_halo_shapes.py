#!/usr/bin/env python3
"""Per-shape timing of the 3x3 conv (forward / data gradient) on the cfg-2 layer shapes with bf16 activations (GPU box).
   python tools/bench_halo_shapes.py [B]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-generation-models_amd")]
import torch
from src.ops import functional as K


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
for H, Ci, Co in [(32, 128, 128), (16, 128, 256), (16, 256, 256), (16, 512, 128), (16, 128, 128), (8, 256, 512), (8, 512, 512), (8, 1024, 256), (8, 256, 256)]:
    x = torch.randn(B, H, H, Ci, device="cuda").bfloat16()
    w = (torch.randn(9 * Co * Ci, device="cuda") * 0.05).bfloat16()
    y = torch.empty(B, H, H, Co, device="cuda", dtype=torch.bfloat16)
    yf = torch.empty(B, H, H, Co, device="cuda")
    fl = 2.0 * B * H * H * Ci * Co * 9
    t = timeit(lambda: K.conv3x3_bf16w(x, w, K=Ci, Nc=Co, flip=False, out=y))
    t2 = timeit(lambda: K.conv3x3_bf16w(x, w, K=Ci, Nc=Co, flip=True, out=yf, accumulate=True))
    print(f"B{B} {H}x{H} {Ci}->{Co}: bf16->bf16 {t*1e6:7.1f} us {fl/t/1e12:6.1f} TF | bf16->fp32 acc {t2*1e6:7.1f} us {fl/t2/1e12:6.1f} TF", flush=True)
