#!/usr/bin/env python3
"""What do the end-of-kernel atomics of the GroupNorm / LayerNorm backward cost?  Each kernel with and without its per-channel outputs."""
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


B = 128
bf = torch.bfloat16
for H, C in [(32, 128), (16, 128), (16, 256), (8, 256), (8, 512)]:
    x = torch.randn(B, H, H, C, device="cuda").to(bf)
    ga = torch.ones(C, device="cuda"); be = torch.zeros(C, device="cuda"); tb = torch.randn(B, C, device="cuda")
    y, st = K.gn_mish_fwd(x, ga, be, temb=tb, out_dtype=bf)
    dg, db, dbias = (torch.zeros(C, device="cuda") for _ in range(3)); dtb = torch.zeros(B, C, device="cuda")
    for name, dout in (("dout bf16", y), ("dout fp32", y.float())):
        t1 = timeit(lambda: K.gn_mish_bwd(x, st, ga, be, dout, dgamma=dg, dbeta=db, dtemb=dtb, dbias=dbias, out_dtype=bf))
        t0 = timeit(lambda: K.gn_mish_bwd(x, st, ga, be, dout, dtemb=dtb, out_dtype=bf))
        tf = timeit(lambda: K.gn_mish_fwd(x, ga, be, temb=tb, out_dtype=bf))
        print(f"GN  {H}x{H} C{C} {name}: bwd {t1:6.1f} us, without atomics {t0:6.1f} us | fwd {tf:6.1f}", flush=True)
    xs = torch.randn(B, H, H, C, device="cuda"); g1 = torch.ones(C, device="cuda"); b1 = torch.zeros(C, device="cuda")
    dxs = torch.zeros(B, H, H, C, device="cuda"); dg1 = torch.zeros(C, device="cuda"); db1 = torch.zeros(C, device="cuda")
    yy = K.chan_layernorm_fwd(xs, g1, b1, out_dtype=bf)
    t1 = timeit(lambda: K.chan_layernorm_bwd(xs, g1, yy, dxs, True, dg1, db1))
    t0 = timeit(lambda: K.chan_layernorm_bwd(xs, g1, yy, dxs, True, None, None))
    print(f"LN  {H}x{H} C{C}: bwd {t1:6.1f} us, without atomics {t0:6.1f} us", flush=True)
