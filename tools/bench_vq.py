#!/usr/bin/env python3
"""Micro-benchmark of the VQ codebook step (mi_vq_nearest_fwd / mi_vq_bwd) with its roofline.
   python tools/bench_vq.py [M D K]   (default: 256 x 32 x 32 rows, D=64, K=512 -- a B=256 VQ-VAE batch)"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
K = importlib.import_module("image-generation-models_amd.src.ops.functional")


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3


def main():
    M, D, Kc = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (256 * 32 * 32, 64, 512)
    torch.manual_seed(0)
    rows = torch.randn(M, D, device="cuda")
    cb = torch.randn(Kc, D, device="cuda")
    t = timeit(lambda: K.vq_nearest(rows, cb))
    idx, zq, _ = K.vq_nearest(rows, cb)
    flops = 2.0 * M * D * Kc
    print(f"vq_nearest  M={M} D={D} K={Kc}: {t*1e6:8.1f} us  {flops/t/1e12:6.1f} TFLOP/s fp32-MFMA (peak ~157)  "
          f"{(2*M*D*4)/t/1e9:7.0f} GB/s algorithmic (z in + q out)")
    dz = torch.empty_like(rows); dcb = torch.zeros_like(cb)
    tb = timeit(lambda: K.vq_backward(rows, cb, idx, 1.0, 0.25, dz=dz, dcodebook=dcb))
    print(f"vq_backward: {tb*1e6:8.1f} us  {(3*M*D*4)/tb/1e9:7.0f} GB/s (z, q in; dz out; + atomics into the codebook)")

    def torch_ref():
        d = torch.cdist(rows, cb)
        i = d.argmin(1)
        q = cb[i]
        return ((rows - q) ** 2).mean(), q
    tt = timeit(torch_ref)
    print(f"torch (cdist + argmin + gather + mse) on the same device: {tt*1e6:8.1f} us  ({tt/t:.1f}x)")


if __name__ == "__main__":
    main()
