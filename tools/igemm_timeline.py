#!/usr/bin/env python3
"""Phase timeline of igemm_fast_kernel workgroups (profiling build: make EXTRA=-DMI_HALO_TIMING)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-generation-models_amd")]
import numpy as np
import torch
from src.ops import functional as K
from src.ops.lib import load_library

lib = load_library()
fn = lib.mi_debug_igemm_ts
fn.argtypes = [ctypes.c_void_p]; fn.restype = ctypes.c_int
B = 128
for IH, OH, Cc, k, T, name in [(32, 16, 128, 3, 0, "Downsample fwd"), (16, 32, 128, 3, 1, "Downsample dgrad"),
                               (16, 32, 128, 4, 1, "Upsample fwd"), (32, 16, 128, 4, 0, "Upsample dgrad"), (16, 8, 256, 4, 0, "Upsample dgrad")]:
    x = torch.randn(B, IH, IH, Cc, device="cuda")
    w = torch.randn(k, k, Cc, Cc, device="cuda") * 0.05
    wb = w.permute(0, 1, 3, 2).contiguous().to(torch.bfloat16).reshape(-1)
    y = torch.empty(B, OH, OH, Cc, device="cuda")
    run = lambda: K.conv_igemm(x, w, kh=k, kw=k, stride=2, pad=1, transposed=bool(T), w_kn=True, K=Cc, Nc=Cc, out_hw=(OH, OH), mode=1, out=y, wb=wb)
    for _ in range(3): run()
    torch.cuda.synchronize()
    buf0 = np.zeros(4 * 4096, dtype=np.uint64)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    buf = np.zeros(4 * 4096, dtype=np.uint64)
    assert fn(buf.ctypes.data) == 0
    ts = buf.reshape(4, 4096).astype(np.int64)
    ok = (ts[0] > 0) & (ts[3] > ts[0]) & (ts[3] - ts[0] < 10**6)
    ts = ts[:, ok]
    t0 = ts[0].min()
    d = np.diff(ts, axis=0) * 10e-3
    print(f"{name:17s} {IH}->{OH} C{Cc} k{k}: kernel {e0.elapsed_time(e1)*1e3:.1f} us, {ts.shape[1]} wgs sampled; mean us: setup+prologue {d[0].mean():.2f} "
          f"main {d[1].mean():.2f} epilogue {d[2].mean():.2f} total {(ts[3]-ts[0]).mean()*10e-3:.2f}; last end {(ts[3].max()-t0)*10e-3:.1f} us")
