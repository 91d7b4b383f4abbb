#!/usr/bin/env python3
"""Micro-benchmark of the conv kernels on the cfg-2 layer shapes (run on the GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-generation-models_amd"), os.path.join(ROOT, "tests")]
import numpy as np
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


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    shapes = [(32, 128, 128), (16, 128, 256), (16, 256, 256), (16, 512, 128), (8, 256, 512), (8, 512, 512), (8, 1024, 256)]
    for H, Ci, Co in shapes:
        x = torch.randn(B, H, H, Ci, device=DEV)
        w = torch.randn(3, 3, Ci, Co, device=DEV) * 0.05
        dy = torch.randn(B, H, H, Co, device=DEV)
        wd = w.to(torch.bfloat16).reshape(-1)
        wf = w.permute(0, 1, 3, 2).contiguous().to(torch.bfloat16).reshape(-1)
        fl = 2.0 * B * H * H * Ci * Co * 9
        y = torch.empty(B, H, H, Co, device=DEV)
        dx = torch.empty(B, H, H, Ci, device=DEV)
        dW = torch.zeros(9 * Ci * Co, device=DEV)
        res = {}
        res["igemm_fwd"] = timeit(lambda: K.conv_igemm(x, w, kh=3, kw=3, stride=1, pad=1, transposed=False, w_kn=True, K=Ci, Nc=Co,
                                                       out_hw=(H, H), mode=1, out=y))
        res["halo_fwd"] = timeit(lambda: K.conv3x3_bf16w(x, wf, K=Ci, Nc=Co, flip=False, out=y))
        res["halo_dgrad"] = timeit(lambda: K.conv3x3_bf16w(dy, wd, K=Co, Nc=Ci, flip=True, out=dx))
        res["wgrad"] = timeit(lambda: K.conv_wgrad(x, dy, dW, kh=3, kw=3, stride=1, pad=1, gather_i=True, Ci=Ci, Cj=Co,
                                                   grid_g=(H, H), grid_d=(H, H), mode=1))
        print(f"B{B} {H}x{H} {Ci}->{Co}: " + "  ".join(f"{k} {v*1e6:7.1f}us {fl/v/1e12:6.1f}TF" for k, v in res.items()), flush=True)
    # 1x1 and elementwise
    for H, Ci, Co in [(32, 128, 384), (32, 128, 128), (16, 256, 384), (8, 512, 384), (8, 128, 512)]:
        x = torch.randn(B, H, H, Ci, device=DEV); w = torch.randn(1, 1, Ci, Co, device=DEV)
        y = torch.empty(B, H, H, Co, device=DEV)
        t = timeit(lambda: K.conv_igemm(x, w, kh=1, kw=1, stride=1, pad=0, transposed=False, w_kn=True, K=Ci, Nc=Co, out_hw=(H, H), mode=1, out=y))
        wf = w.permute(0, 1, 3, 2).contiguous().to(torch.bfloat16).reshape(-1)
        t2 = timeit(lambda: K.conv3x3_bf16w(x, wf, K=Ci, Nc=Co, flip=False, ksize=1, out=y))
        dW = torch.zeros(Ci * Co, device=DEV)
        t3 = timeit(lambda: K.conv_wgrad(x, y, dW, kh=1, kw=1, stride=1, pad=0, gather_i=True, Ci=Ci, Cj=Co, grid_g=(H, H), grid_d=(H, H), mode=1))
        fl = 2.0 * B * H * H * Ci * Co
        print(f"B{B} 1x1 {H}x{H} {Ci}->{Co}: igemm {t*1e6:7.1f}us {fl/t/1e12:6.1f}TF | tile {t2*1e6:7.1f}us {fl/t2/1e12:6.1f}TF {(x.numel()+y.numel())*4/t2/1e9:6.0f}GB/s | wgrad {t3*1e6:7.1f}us {fl/t3/1e12:6.1f}TF", flush=True)
    for H, C in [(32, 128), (16, 256), (8, 512)]:
        x = torch.randn(B, H, H, C, device=DEV); ga = torch.ones(C, device=DEV); be = torch.zeros(C, device=DEV)
        t = timeit(lambda: K.gn_mish_fwd(x, ga, be))
        print(f"B{B} gn_mish_fwd {H}x{H} C{C}: {t*1e6:7.1f}us {x.numel()*8/t/1e9:7.0f}GB/s", flush=True)
        y, st = K.gn_mish_fwd(x, ga, be)
        t = timeit(lambda: K.gn_mish_bwd(x, st, ga, be, y))
        print(f"B{B} gn_mish_bwd {H}x{H} C{C}: {t*1e6:7.1f}us {x.numel()*12/t/1e9:7.0f}GB/s", flush=True)


if __name__ == "__main__":
    main()
