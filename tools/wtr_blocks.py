#!/usr/bin/env python3
"""LDS-DMA weight-gradient kernel: time per launch (20 back-to-back launches between one event pair, so the queue never runs dry)
of the contraction kernel alone, the reduce alone and both, for several workgroup counts; and the register-staged kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-generation-models_amd")]
import torch
from src.ops import functional as K
from src.ops.lib import load_library
lib = load_library()
B = int(os.environ.get("B", 128))


def timed(run, n=20):
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for H, Ci, Cj in [(32, 128, 128), (16, 256, 256), (8, 512, 512)]:
    x = torch.randn(B, H, H, Ci, device="cuda").bfloat16(); dy = torch.randn(B, H, H, Cj, device="cuda").bfloat16()
    dW = torch.zeros(9 * Ci * Cj, device="cuda")
    run = lambda: K.conv_wgrad(x, dy, dW, kh=3, kw=3, stride=1, pad=1, gather_i=True, Ci=Ci, Cj=Cj, grid_g=(H, H), grid_d=(H, H), mode=1)
    K.USE_WGRAD_TR = False
    old = timed(run)
    K.USE_WGRAD_TR = True
    line = f"{H}x{H} {Ci}->{Cj}: old kernel+reduce {old:.1f}us |"
    for blocks in [int(b) for b in os.environ.get("BLOCKS", "256,128,64,32").split(",")]:
        lib.mi_debug_wgrad_tr_blocks(blocks)
        both = timed(run)
        lib.mi_debug_wgrad_tr_phase(1); main = timed(run)
        lib.mi_debug_wgrad_tr_phase(2); red = timed(run)
        lib.mi_debug_wgrad_tr_phase(0)
        line += f" {blocks} wgs: {main:.1f} + {red:.1f} = {both:.1f}us |"
    lib.mi_debug_wgrad_tr_blocks(0)
    print(line, flush=True)
