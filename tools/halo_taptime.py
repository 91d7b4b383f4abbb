#!/usr/bin/env python3
"""Where a tap of the 3x3 conv's main loop goes, per wave (needs: make -C image-generation-models_amd clean all EXTRA=-DMI_HALO_TAPTIME):
shader-clock cycles between barriers (work: loads issued, MFMAs issued, LDS stores issued) and at the barrier (own LDS operations
drained + waiting for the other waves).   python tools/halo_taptime.py H Ci Co"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-generation-models_amd")]
import numpy as np
import torch
from src.ops import functional as K
from src.ops.lib import load_library

lib = load_library()
fn = lib.mi_debug_halo_tap
fn.argtypes = [ctypes.c_void_p]; fn.restype = ctypes.c_int
B = 128
H, Ci, Co = (int(v) for v in sys.argv[1:4])        # one shape per process: the device buffer keeps earlier launches' entries
for _ in range(1):
    x = torch.randn(B, H, H, Ci, device="cuda").bfloat16()
    w = (torch.randn(9 * Co * Ci, device="cuda") * 0.05).to(torch.bfloat16)
    y = torch.empty(B, H, H, Co, device="cuda", dtype=torch.bfloat16)
    for _ in range(200):          # steady state: clocks and caches as in a training step
        K.conv3x3_bf16w(x, w, K=Ci, Nc=Co, flip=False, out=y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); K.conv3x3_bf16w(x, w, K=Ci, Nc=Co, flip=False, out=y); e1.record()
    torch.cuda.synchronize()
    buf = np.zeros(4096 * 8 * 5, dtype=np.uint64)
    assert fn(buf.ctypes.data) == 0
    t = buf.reshape(4096, 8, 5).astype(np.float64)
    ok = t[:, 0, 2] > 0
    t = t[ok]
    n = t[:, :, 2]
    print(f"B{B} {H}x{H} {Ci}->{Co}: kernel {e0.elapsed_time(e1)*1e3:.1f} us, {t.shape[0]} wgs x 8 waves, {n.mean():.0f} taps; per tap: "
          f"work {(t[:, :, 0] / n).mean():.0f} cycles (per wave min {(t[:, :, 0] / n).mean(0).min():.0f} max {(t[:, :, 0] / n).mean(0).max():.0f}), "
          f"barrier {(t[:, :, 1] / n).mean():.0f} cycles (min {(t[:, :, 1] / n).mean(0).min():.0f} max {(t[:, :, 1] / n).mean(0).max():.0f}); "
          f"of the work, barrier exit -> LDS stores issued {(t[:, :, 4] / n).mean():.0f} cycles; "
          f"main loop {t[:, :, 3].mean() * 10e-3:.2f} us by the 100 MHz clock => {((t[:, :, 0] + t[:, :, 1]) / (t[:, :, 3] * 10e-3)).mean():.0f} counter ticks per us", flush=True)
    buf[:] = 0
