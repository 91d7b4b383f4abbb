#!/usr/bin/env python3
"""GroupNorm + Mish forward: the two-phase kernel against the streaming apply fed by epilogue sums (cfg-2 shapes, B = 128, bf16 x)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-generation-models_amd")]
import torch
from src.ops import functional as K
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = "cuda"
def timeit(fn, n=20):
    """n launches captured into ONE hipGraph and replayed: device time per launch, free of the Python launch overhead (~12 us per call)"""
    for _ in range(3): fn()
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
for H, C in [(32, 128), (16, 256), (16, 128), (8, 512), (8, 256)]:
    x = torch.randn(B, H, H, C, device=dev).bfloat16()
    ga = torch.ones(C, device=dev); be = torch.zeros(C, device=dev); tb = torch.randn(B, C, device=dev)
    res = torch.randn(B, H, H, C, device=dev)
    xs = x.double().view(B, H * H, C // 16, 16)
    sums = K.gn_sums_encode(torch.stack([xs.sum((1, 3)), (xs * xs).sum((1, 3))], dim=-1))
    r = {}
    r["2ph bf16"] = timeit(lambda: K.gn_mish_fwd(x, ga, be, temb=tb, out_dtype=torch.bfloat16))
    r["apply bf16"] = timeit(lambda: K.gn_mish_apply_sums(x, sums, ga, be, temb=tb, out_dtype=torch.bfloat16))
    r["2ph f32+res+copy"] = timeit(lambda: K.gn_mish_fwd(x, ga, be, residual=res, want16=True))
    r["apply f32+res+copy"] = timeit(lambda: K.gn_mish_apply_sums(x, sums, ga, be, residual=res, want16=True))
    r["2ph f32+res"] = timeit(lambda: K.gn_mish_fwd(x, ga, be, residual=res))
    r["apply f32+res"] = timeit(lambda: K.gn_mish_apply_sums(x, sums, ga, be, residual=res))
    print(f"B{B} {H}x{H} C{C}: " + "  ".join(f"{k} {v:5.1f}" for k, v in r.items()), flush=True)
