#!/usr/bin/env python3
"""conv_pw's split-K launches on the shapes that leave CUs without a workgroup: unsplit against split 2 / 4 (mi_debug_conv_pw_splitk), 20 back-to-back
launches per timing, interleaved rounds, median.  B=64 (sampler) by default; CFG3=1: B=32 shapes.  a library built with -DMI_PW_SK_ABL: no exchange (upper bound, wrong results)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-generation-models_amd")]
import torch
from src.ops import functional as K
B = int(os.environ.get("B", 64))
lib = K.load_library()


def timed(run, n=20):
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


SHAPES = [(8, 512, 512), (8, 1024, 512), (16, 256, 256), (16, 512, 256)]
for H, Ci, Co in SHAPES:
    x = torch.randn(B, H, H, Ci, device="cuda").bfloat16()
    w = torch.randn(3, 3, Ci, Co, device="cuda") * 0.05
    table, nent, tiles = K.pack_table([(0, 9, Ci, Co)], "cuda")
    wd, wf, wdq, wfq = (torch.zeros(w.numel(), device="cuda", dtype=torch.bfloat16) for _ in range(4))
    K.pack_weights_bf16(table, nent, tiles, w.reshape(-1), wd, wf, wdq, wfq)
    y16 = torch.empty(B, H, H, Co, device="cuda", dtype=torch.bfloat16)
    fl = 2.0 * B * H * H * Ci * Co * 9
    line = f"B={B} {H}x{H} {Ci}->{Co}:"
    res = {}
    for rnd in range(5):
        for mode, tile in ((0, 0), (2, 128), (4, 128), (2, 64), (4, 64)):
            lib.mi_debug_conv_pw_splitk(mode); lib.mi_debug_conv_pw_tile(tile); K._QUERY_CACHE.clear()
            d = K.MiConvDesc(N=B, IH=H, IW=H, OH=H, OW=H, K=Ci, Nc=Co, KH=3, KW=3, stride=1, pad=1, transposed=0, w_kn=0, mode=K.MODE_BF16, K1=Ci, ldx=Ci, ldx2=0, ldy=Co, ldr=0, accumulate=0)
            import ctypes
            q = lib.mi_conv3x3_pw_splitk(ctypes.byref(d), 0, 0)
            res.setdefault((mode, tile, q & 15, q >> 4), []).append(timed(lambda: K.conv3x3_bf16w(x, wf, K=Ci, Nc=Co, flip=False, out=y16, wq=wfq)))
    lib.mi_debug_conv_pw_splitk(1); lib.mi_debug_conv_pw_tile(0); K._QUERY_CACHE.clear()
    for k, v in res.items():
        m = sorted(v)[len(v) // 2]
        line += f"  [split {k[2]} tile {k[3]}] {m:.1f}us {fl / m / 1e6:.0f}TF |"
    print(line, flush=True)
