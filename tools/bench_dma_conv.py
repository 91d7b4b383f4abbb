#!/usr/bin/env python3
"""3x3 conv forward / data gradient on the cfg-2 shapes, bf16-stored activations: register-staged halo kernel (conv3x3_halo.hip)
against the LDS-DMA kernel (conv_dma.hip); 20 back-to-back launches per timing, interleaved rounds, median."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-generation-models_amd")]
import torch
from src.ops import functional as K
B = int(os.environ.get("B", 128))
K.CONV_AUTO = False            # "halo" below means the halo kernel, not the per-shape pick


def timed(run, n=20):
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for H, Ci, Co in [(32, 128, 128), (16, 128, 256), (16, 256, 256), (16, 512, 128), (8, 256, 512), (8, 512, 512), (8, 1024, 256)]:
    x = torch.randn(B, H, H, Ci, device="cuda").bfloat16()
    wf = (torch.randn(9 * Co * Ci, device="cuda") * 0.05).bfloat16()
    y16 = torch.empty(B, H, H, Co, device="cuda", dtype=torch.bfloat16)
    y32 = torch.empty(B, H, H, Co, device="cuda")
    fl = 2.0 * B * H * H * Ci * Co * 9
    line = f"{H}x{H} {Ci}->{Co}:"
    for name, out in (("bf16 out", y16), ("fp32 out", y32)):
        res = {"halo": [], "dma": [], "shift": []}
        for rnd in range(3):
            for which in res:
                K.USE_CONV_DMA, K.USE_CONV_SHIFT = which == "dma", which == "shift"
                res[which].append(timed(lambda: K.conv3x3_bf16w(x, wf, K=Ci, Nc=Co, flip=False, out=out)))
        line += f"  [{name}]" + " |".join(f" {k} {sorted(v)[1]:.1f}us {fl / sorted(v)[1] / 1e6:.0f}TF" for k, v in res.items())
    print(line, flush=True)
