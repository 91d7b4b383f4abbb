#!/usr/bin/env python3
"""Micro-benchmark of the stride-2 / transposed convolutions of cfg 2 (Downsample, Upsample and their data gradients)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-generation-models_amd")]
import torch
from src.ops import functional as K

DEV = "cuda"


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
# (IH, OH, C, k, transposed)
for IH, OH, Cc, k, T, name in [(32, 16, 128, 3, 0, "Downsample fwd"), (16, 8, 256, 3, 0, "Downsample fwd"),
                               (16, 32, 128, 3, 1, "Downsample dgrad"), (8, 16, 256, 3, 1, "Downsample dgrad"),
                               (16, 32, 128, 4, 1, "Upsample fwd"), (8, 16, 256, 4, 1, "Upsample fwd"),
                               (32, 16, 128, 4, 0, "Upsample dgrad"), (16, 8, 256, 4, 0, "Upsample dgrad")]:
    x = torch.randn(B, IH, IH, Cc, device=DEV)
    w = torch.randn(k, k, Cc, Cc, device=DEV) * 0.05
    wb = w.permute(0, 1, 3, 2).contiguous().to(torch.bfloat16).reshape(-1)
    y = torch.empty(B, OH, OH, Cc, device=DEV)
    fl = 2.0 * B * OH * OH * Cc * Cc * (k * k / 4 if T else k * k)
    t = timeit(lambda: K.conv_igemm(x, w, kh=k, kw=k, stride=2, pad=1, transposed=bool(T), w_kn=True, K=Cc, Nc=Cc,
                                    out_hw=(OH, OH), mode=1, out=y, wb=wb))
    x16 = x.bfloat16()
    t16 = timeit(lambda: K.conv_igemm(x16, w, kh=k, kw=k, stride=2, pad=1, transposed=bool(T), w_kn=True, K=Cc, Nc=Cc,
                                      out_hw=(OH, OH), mode=1, out=y, wb=wb))
    # the tap-gather kernel (round 4): fragment-order weights
    table, nent, tiles = K.pack_table([(0, k * k, Cc, Cc)], DEV)
    WQ = [torch.zeros(w.numel(), device=DEV, dtype=torch.bfloat16) for _ in range(4)]
    K.pack_weights_bf16(table, nent, tiles, w.reshape(-1), *WQ)
    tg = timeit(lambda: K.conv_gt(x16, WQ[3], kh=k, kw=k, stride=2, pad=1, transposed=bool(T), K=Cc, Nc=Cc, out_hw=(OH, OH), out=y))
    yr = K.conv_igemm(x16, w, kh=k, kw=k, stride=2, pad=1, transposed=bool(T), w_kn=True, K=Cc, Nc=Cc, out_hw=(OH, OH), mode=1, wb=wb)
    yg = K.conv_gt(x16, WQ[3], kh=k, kw=k, stride=2, pad=1, transposed=bool(T), K=Cc, Nc=Cc, out_hw=(OH, OH))
    err = float((yr - yg).norm() / yr.norm())
    print(f"B{B} {name:17s} {IH}x{IH}->{OH}x{OH} C{Cc} k{k}: igemm fp32-in {t*1e6:6.1f} us | bf16-in {t16*1e6:6.1f} us {fl/t16/1e12:6.1f} TF | "
          f"tap-gather {tg*1e6:6.1f} us {fl/tg/1e12:6.1f} TF  (rel diff {err:.1e})", flush=True)
