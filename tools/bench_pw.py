#!/usr/bin/env python3
"""3x3 conv forward on the cfg-2 / cfg-3 shapes, bf16-stored activations: round 2's register-staged halo kernel (no fragment-order weights passed) against the
private-weight-stream kernel (conv_pw.hip); 20 back-to-back launches per timing, interleaved rounds, median.  Random operands."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-generation-models_amd")]
import torch
from src.ops import functional as K
B = int(os.environ.get("B", 128))
if os.environ.get("PW_TILE"):       # force conv_pw's pixel tile (64 / 128)
    K.load_library().mi_debug_conv_pw_tile(int(os.environ["PW_TILE"])); K.PW_MIN_TILES = 0


def timed(run, n=20):
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


SHAPES = [(32, 128, 128), (16, 128, 128), (16, 128, 256), (16, 256, 256), (16, 512, 128), (8, 256, 256), (8, 256, 512), (8, 512, 512),
          (8, 1024, 256)]
if os.environ.get("SHORT"):
    SHAPES = [(32, 128, 128), (16, 256, 256), (16, 512, 128), (8, 512, 512), (8, 1024, 256)]
if os.environ.get("CFG3"):
    SHAPES = [(32, 128, 128), (32, 256, 64), (16, 128, 128), (16, 256, 256), (16, 512, 128), (8, 256, 256), (8, 512, 512), (8, 1024, 256)]
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
    for name, out in (("bf16 out", y16), ("fp32 out", y32)):
        res = {"r2": [], "pw": []}
        for rnd in range(5):
            for which in res:
                wq = wfq if which == "pw" else None
                res[which].append(timed(lambda: K.conv3x3_bf16w(x, wf, K=Ci, Nc=Co, flip=False, out=out, wq=wq)))
        med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
        line += f"  [{name}]" + " |".join(f" {k} {m:.1f}us {fl / m / 1e6:.0f}TF" for k, m in med.items())
    ya = K.conv3x3_bf16w(x, wf, K=Ci, Nc=Co, flip=False, out_dtype=torch.float32)
    yb = K.conv3x3_bf16w(x, wf, K=Ci, Nc=Co, flip=False, out_dtype=torch.float32, wq=wfq)
    err = float((ya - yb).norm() / ya.norm())
    print(line + f"  rel diff {err:.1e}", flush=True)
