#!/usr/bin/env python3
"""Micro-benchmark of GroupNorm+Mish forward / backward on the cfg-2 shapes, fp32 and bf16 storage."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-generation-models_amd")]
import torch
from src.ops import functional as K


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
REP = int(sys.argv[2]) if len(sys.argv) > 2 else 1          # the whole table REP times (drift inside one process)
for dt in [torch.float32, torch.bfloat16] * REP:
    for H, C in [(32, 128), (16, 256), (8, 512)]:
        x = torch.randn(B, H, H, C, device="cuda").to(dt)
        ga = torch.ones(C, device="cuda"); be = torch.zeros(C, device="cuda")
        tb = torch.randn(B, C, device="cuda")
        y, st = K.gn_mish_fwd(x, ga, be, temb=tb, out_dtype=dt)
        tf = timeit(lambda: K.gn_mish_fwd(x, ga, be, temb=tb, out_dtype=dt))
        dg, db, dbias = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        dtb = torch.zeros(B, C, device="cuda")
        tbw = timeit(lambda: K.gn_mish_bwd(x, st, ga, be, y, dgamma=dg, dbeta=db, dtemb=dtb, dbias=dbias, out_dtype=dt))
        el = x.numel(); bpe = 4 if dt == torch.float32 else 2
        print(f"B{B} {str(dt)[6:]:9s} {H}x{H} C{C}: fwd {tf*1e6:6.1f} us {el*2*bpe/tf/1e9:6.0f} GB/s | bwd {tbw*1e6:6.1f} us {el*3*bpe/tbw/1e9:6.0f} GB/s", flush=True)
