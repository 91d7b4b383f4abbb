#!/usr/bin/env python3
"""conv_pw: 128-pixel tiles (two workgroups per CU) against 256-pixel tiles (round 5: one workgroup per CU, a weight fragment feeds
eight MFMAs) on the cfg-2 shapes that offer >= 128 tiles of 256 pixels; interleaved rounds, median, random operands."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-generation-models_amd")]
import torch
from src.ops import functional as K
B = int(os.environ.get("B", 128))
lib = K.load_library()
K.PW_MIN_TILES = 0
TILES = [int(t) for t in os.environ.get("TILES", "128,256").split(",")]


def timed(run, n=20):
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


SHAPES = [(32, 128, 128), (16, 256, 256), (16, 128, 256), (16, 512, 128), (16, 128, 128), (32, 256, 128)]
for H, Ci, Co in SHAPES:
    x = torch.randn(B, H, H, Ci, device="cuda").bfloat16()
    w = torch.randn(3, 3, Ci, Co, device="cuda") * 0.05
    table, nent, tiles = K.pack_table([(0, 9, Ci, Co)], "cuda")
    wd, wf, wdq, wfq = (torch.zeros(w.numel(), device="cuda", dtype=torch.bfloat16) for _ in range(4))
    K.pack_weights_bf16(table, nent, tiles, w.reshape(-1), wd, wf, wdq, wfq)
    y16 = torch.empty(B, H, H, Co, device="cuda", dtype=torch.bfloat16)
    y32 = torch.empty(B, H, H, Co, device="cuda")
    fl = 2.0 * B * H * H * Ci * Co * 9
    line = f"{H}x{H} {Ci}->{Co}:"
    outs = {}
    for name, out in (("bf16 out", y16), ("fp32 out", y32)):
        res = {t: [] for t in TILES}
        for rnd in range(5):
            for t in TILES:
                lib.mi_debug_conv_pw_tile(t); K._QUERY_CACHE.clear()
                res[t].append(timed(lambda: K.conv3x3_bf16w(x, wf, K=Ci, Nc=Co, flip=False, out=out, wq=wfq)))
        med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
        line += f"  [{name}]" + " |".join(f" pt{k} {m:.1f}us {fl / m / 1e6:.0f}TF" for k, m in med.items())
    ys = []
    for t in TILES:
        lib.mi_debug_conv_pw_tile(t); K._QUERY_CACHE.clear()
        ys.append(K.conv3x3_bf16w(x, wf, K=Ci, Nc=Co, flip=False, out_dtype=torch.float32, wq=wfq).clone())
    err = max(float((ys[0] - y).abs().max()) for y in ys[1:]) if len(ys) > 1 else 0.0
    print(line + f"  max abs diff {err:.1e}", flush=True)
lib.mi_debug_conv_pw_tile(0)
