#!/usr/bin/env python3
"""The eight-layer batched 3x3 weight-gradient launch of cfg 2's backward (4x 128->128 @32x32, 3x 256->256 + 512->128 @16x16) and the
single-layer launches, device time from a replayed hipGraph (contraction + reduce).  MI_DDPM_LIB=... for A/B builds."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-generation-models_amd")]
import torch
from src.ops import functional as K
B = int(os.environ.get("B", 128)); DEV = "cuda"
def graph_time(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1e3)
    return sorted(ts)[2]
def group(shapes):
    Q = [(torch.randn(B, h, h, ci, device=DEV).bfloat16(), torch.randn(B, h, h, co, device=DEV).bfloat16(), torch.zeros(9 * ci * co, device=DEV), h, ci, co) for h, ci, co in shapes]
    def run():
        q = K.WgradQueue(group=8)
        for (xx, dd, ww, h, ci, co) in Q:
            q.push(xx, dd, ww, Ci=ci, Cj=co, hw=(h, h), mode=1)
        q.flush()
    fl = sum(2.0 * B * h * h * ci * co * 9 for h, ci, co in shapes)
    t = graph_time(run)
    return t, fl / t / 1e6
tag = os.environ.get("MI_DDPM_LIB", "default")[-22:]
t, tf = group([(32, 128, 128)] * 4 + [(16, 256, 256)] * 2 + [(16, 512, 128), (16, 256, 256)])
line = f"{tag}: 8-layer group {t:.1f} us {tf:.0f} TFLOP/s"
for name, shp in (("8x 32x32 128", [(32, 128, 128)] * 8), ("8x 16x16 256", [(16, 256, 256)] * 8), ("8x 8x8 512", [(8, 512, 512)] * 8)):
    t, tf = group(shp)
    line += f" | {name}: {t:.1f} us {tf:.0f}"
print(line, flush=True)
