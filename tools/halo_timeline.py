#!/usr/bin/env python3
"""Phase timeline of conv3x3_halo workgroups (needs a profiling build: make -C image-generation-models_amd clean all EXTRA=-DMI_HALO_TIMING).
Prints, per layer shape, the mean time a workgroup spends in setup / prologue / main loop / epilogue."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-generation-models_amd")]
import numpy as np
import torch
from src.ops import functional as K
from src.ops.lib import load_library

lib = load_library()
fn = lib.mi_debug_halo_ts
fn.argtypes = [ctypes.c_void_p]; fn.restype = ctypes.c_int
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
DT = torch.bfloat16 if (len(sys.argv) > 2 and sys.argv[2] == "bf16") else torch.float32
for H, Ci, Co in [(32, 128, 128), (32, 256, 128), (16, 256, 256), (16, 512, 128), (8, 512, 512)]:
    x = torch.randn(B, H, H, Ci, device="cuda").to(DT)
    w = (torch.randn(3, 3, Co, Ci, device="cuda") * 0.05).to(torch.bfloat16).reshape(-1)
    y = torch.empty(B, H, H, Co, device="cuda", dtype=DT)
    for _ in range(3):
        K.conv3x3_bf16w(x, w, K=Ci, Nc=Co, flip=False, out=y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); K.conv3x3_bf16w(x, w, K=Ci, Nc=Co, flip=False, out=y); e1.record()
    torch.cuda.synchronize()
    buf = np.zeros(5 * 4096, dtype=np.uint64)
    assert fn(buf.ctypes.data) == 0
    ts = buf.reshape(5, 4096).astype(np.int64)
    nb = min(4096, (B * H * H + 255) // 256)        # x-blocks (blockIdx.y == 0 only)
    ts = ts[:, :nb]
    ok = ts[0] > 0
    ts = ts[:, ok]
    t0 = ts[0].min()
    d = np.diff(ts, axis=0) * 10e-3                 # us
    print(f"B{B} {H}x{H} {Ci}->{Co}: kernel {e0.elapsed_time(e1)*1e3:.1f} us, {ts.shape[1]} wgs; "
          f"mean us: setup {d[0].mean():.2f} prologue {d[1].mean():.2f} main {d[2].mean():.2f} epilogue {d[3].mean():.2f} "
          f"total {(ts[4]-ts[0]).mean()*10e-3:.2f}; start spread {(ts[0].max()-t0)*10e-3:.1f} us, last end {(ts[4].max()-t0)*10e-3:.1f} us")
