#!/usr/bin/env python3
"""conv_pw reading fp32 activations (the sampler's block1 convs) beside the bf16-input launch, sampler batch (B = 64 by default)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-generation-models_amd")]
import torch
from src.ops import functional as K
B = int(os.environ.get("B", 64))
K.PW_MIN_TILES = 0


def timed(run, n=20):
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for H, Ci, Co in [(32, 128, 128), (16, 128, 256), (16, 256, 256), (8, 256, 512), (8, 512, 512)]:
    x = torch.randn(B, H, H, Ci, device="cuda")
    w = torch.randn(3, 3, Ci, Co, device="cuda") * 0.05
    table, nent, tiles = K.pack_table([(0, 9, Ci, Co)], "cuda")
    wd, wf, wdq, wfq = (torch.zeros(w.numel(), device="cuda", dtype=torch.bfloat16) for _ in range(4))
    K.pack_weights_bf16(table, nent, tiles, w.reshape(-1), wd, wf, wdq, wfq)
    y16 = torch.empty(B, H, H, Co, device="cuda", dtype=torch.bfloat16)
    x16 = x.bfloat16()
    res = {"x32": [], "x16": []}
    for rnd in range(5):
        res["x32"].append(timed(lambda: K.conv3x3_bf16w(x, wf, K=Ci, Nc=Co, flip=False, out=y16, wq=wfq)))
        res["x16"].append(timed(lambda: K.conv3x3_bf16w(x16, wf, K=Ci, Nc=Co, flip=False, out=y16, wq=wfq)))
    med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
    print(f"B{B} {H}x{H} {Ci}->{Co}: fp32 in {med['x32']:.1f} us | bf16 in {med['x16']:.1f} us", flush=True)
