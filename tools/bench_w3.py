#!/usr/bin/env python3
"""3x3 weight gradient on the cfg-2 shapes, bf16-stored operands: the register-staged image-major kernel (wgrad3x3.hip) against the
LDS-DMA + transposing-read kernel (wgrad_tr.hip), contraction and partial-tile reduce timed separately (HIP events, interleaved
rounds in one process, median).  For ablation builds: MI_DDPM_LIB=..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-generation-models_amd")]
import torch
from src.ops import functional as K
B = int(os.environ.get("B", 128))
SHAPES = [(32, 128, 128), (16, 256, 256), (8, 512, 512), (16, 128, 256), (8, 256, 512), (16, 512, 128), (8, 1024, 256)]
for H, Ci, Cj in SHAPES:
    x = torch.randn(B, H, H, Ci, device="cuda").bfloat16(); dy = torch.randn(B, H, H, Cj, device="cuda").bfloat16()
    dW = torch.zeros(9 * Ci * Cj, device="cuda")
    run = lambda: K.conv_wgrad(x, dy, dW, kh=3, kw=3, stride=1, pad=1, gather_i=True, Ci=Ci, Cj=Cj, grid_g=(H, H), grid_d=(H, H), mode=1)
    res = {}
    for rnd in range(6):
        for tr in (False, True):
            K.USE_WGRAD_TR = tr
            for _ in range(2): run()
            K.PROBE = []
            for _ in range(5): run()
            torch.cuda.synchronize()
            for sym, fl, e0, e1, desc, nb in K.PROBE:
                res.setdefault((tr, sym.split("<")[0]), []).append(e0.elapsed_time(e1) * 1e3)
            K.PROBE = None
    fl = 2.0 * B * H * H * Ci * Cj * 9
    line = f"{H}x{H} Ci{Ci} Cj{Cj}:"
    for tr in (False, True):
        tot = 0.0
        for (t_, name), v in sorted(res.items()):
            if t_ == tr:
                v = sorted(v); med = v[len(v) // 2]; tot += med
                line += f"  {name} {med:.1f}us"
        line += f"  [{'new' if tr else 'old'} total {tot:.1f}us = {fl / tot / 1e6:.0f} TFLOP/s]"
    print(line, flush=True)
