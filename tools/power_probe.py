#!/usr/bin/env python3
"""Sustained loop of one 3x3 conv shape while rocm-smi samples clocks and power (GPU box).  python tools/power_probe.py H Ci Co [seconds]"""
import os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-generation-models_amd")]
import torch
from src.ops import functional as K

H, Ci, Co = (int(v) for v in sys.argv[1:4])
secs = float(sys.argv[4]) if len(sys.argv) > 4 else 4.0
B = 128
x = torch.randn(B, H, H, Ci, device="cuda").bfloat16()
w = (torch.randn(9 * Co * Ci, device="cuda") * 0.05).bfloat16()
y = torch.empty(B, H, H, Co, device="cuda", dtype=torch.bfloat16)
out = []


def sample():
    time.sleep(secs * 0.4)
    for _ in range(3):
        r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp"], capture_output=True, text=True).stdout
        out.append([ln for ln in r.splitlines() if any(k in ln for k in ("sclk", "mclk", "Power", "junction", "fclk"))])
        time.sleep(secs * 0.15)


th = threading.Thread(target=sample); th.start()
t0 = time.time(); n = 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
while time.time() - t0 < secs:
    for _ in range(200):
        K.conv3x3_bf16w(x, w, K=Ci, Nc=Co, flip=False, out=y)
    n += 200
    torch.cuda.synchronize()
e1.record(); torch.cuda.synchronize()
th.join()
us = e0.elapsed_time(e1) * 1e3 / n
print(f"B{B} {H}x{H} {Ci}->{Co}: {us:.1f} us per launch over {n} launches, {2.0 * B * H * H * Ci * Co * 9 / us / 1e6:.0f} TFLOP/s")
for s in out[-1:]:
    print("   ", " | ".join(v.strip().replace("GPU[0]\t\t: ", "") for v in s if "sclk" in v or "Power (W)" in v))
