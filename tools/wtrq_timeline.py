#!/usr/bin/env python3
"""Per-workgroup phase timeline of the batched (eight layers, one launch) 3x3 weight gradient (profiling build: wgrad_tr.hip with
-DMI_WTR_TIMING, MI_DDPM_LIB=.../libmi_wtr_timing.so): when each workgroup starts, how long its loop runs, when it ends."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-generation-models_amd")]
import numpy as np
import torch
from src.ops import functional as K
from src.ops.lib import load_library
lib = load_library()
fn = lib.mi_debug_wtr_ts; fn.argtypes = [ctypes.c_void_p]; fn.restype = ctypes.c_int
B = 128; DEV = "cuda"
Q8 = []
for (h, ci, co) in [(32, 128, 128)] * 4 + [(16, 256, 256)] * 2 + [(16, 512, 128), (16, 256, 256)]:
    Q8.append((torch.randn(B, h, h, ci, device=DEV).bfloat16(), torch.randn(B, h, h, co, device=DEV).bfloat16(), torch.zeros(9 * ci * co, device=DEV), h, ci, co))
def run():
    q = K.WgradQueue(group=8)
    for (xx, dd, ww, h, ci, co) in Q8:
        q.push(xx, dd, ww, Ci=ci, Cj=co, hw=(h, h), mode=1)
    q.flush()
lib.mi_debug_wgrad_tr_phase(1)
for _ in range(5): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
K.PROBE = []
for _ in range(5): run()
torch.cuda.synchronize()
kus = sorted(ev0.elapsed_time(ev1) * 1e3 for _s, _f, ev0, ev1, _d, _n in K.PROBE if "wgrad_tr" in _s and "reduce" not in _s)
K.PROBE = None
print("wgrad_tr_kernel by HIP events around the launch (5 runs, us):", [round(v, 1) for v in kus])
lib.mi_debug_wgrad_tr_phase(0)
buf = np.zeros(6 * 1024, dtype=np.uint64)
assert fn(buf.ctypes.data) == 0
raw5 = buf.reshape(6, 1024)[5].copy()
ts = buf.reshape(6, 1024)[:5].astype(np.int64)
n = int((ts[4] > 0).sum())
ts = ts[:, :n]
t0 = ts[0].min()
us = (ts - t0) * 0.01
print(f"batched launch: {e0.elapsed_time(e1)*1e3:.1f} us, {n} workgroups")
print(f"  start: min {us[0].min():.1f} max {us[0].max():.1f};  loop begins {us[2].mean():.1f} (mean);  loop us: min {(us[3]-us[2]).min():.1f} mean {(us[3]-us[2]).mean():.1f} max {(us[3]-us[2]).max():.1f}")
print(f"  end: first {us[4].min():.1f} mean {us[4].mean():.1f} last {us[4].max():.1f};  stores us mean {(us[4]-us[3]).mean():.1f} max {(us[4]-us[3]).max():.1f}")
end = np.sort(us[4])
print("  end-time deciles:", [round(float(end[int(q * (n - 1))]), 1) for q in np.linspace(0, 1, 11)])
loop = us[3] - us[2]
for r in range(0, n, 32):
    print(f"  wg {r:3d}..: loop us", " ".join(f"{v:4.0f}" for v in loop[r:r + 32]))
fs = lib.mi_debug_wtr_steps; fs.argtypes = [ctypes.c_void_p]; fs.restype = ctypes.c_int
nst = np.zeros(1024, dtype=np.uint64); assert fs(nst.ctypes.data) == 0
wait = (raw5[:n] >> np.uint64(40)).astype(np.float64); work = (raw5[:n] & np.uint64(0xffffffffff)).astype(np.float64); st = nst[:n].astype(np.float64)
print(f"  per step (wave 0 of each workgroup, shader cycles): wait + barrier {np.mean(wait / st):.0f}, the rest {np.mean(work / st):.0f}; steps per workgroup {st.min():.0f}..{st.max():.0f}; MFMA floor per step 36 x 32 x 2 waves per SIMD = 2304")
for r in range(0, n, 64):
    print(f"  wg {r:3d}..: cycles/step wait", " ".join(f"{v:4.0f}" for v in (wait / st)[r:r + 16]), "| work", " ".join(f"{v:4.0f}" for v in (work / st)[r:r + 16]))
print("  mean loop by XCD (wg & 7):", [round(float(loop[x::8].mean()), 1) for x in range(8)])
