#!/usr/bin/env python3
"""bf16-output 1x1 convs of cfg 2 through conv1x1_pw (to_qkv shapes), B = 128."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-generation-models_amd")]
import torch
from src.ops import functional as K
def timed(run, n=30):
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (H, Ci, Co) in [(32, 128, 384), (16, 128, 384), (16, 256, 384), (8, 512, 384)]:
    x = torch.randn(128, H, H, Ci, device="cuda").bfloat16()
    w1 = torch.randn(1, 1, Ci, Co, device="cuda") * 0.05
    table, nent, tiles = K.pack_table([(0, 1, Ci, Co)], "cuda")
    W1 = [torch.zeros(w1.numel(), device="cuda", dtype=torch.bfloat16) for _ in range(4)]
    K.pack_weights_bf16(table, nent, tiles, w1.reshape(-1), *W1)
    ts = sorted(timed(lambda: K.conv3x3_bf16w(x, W1[1], K=Ci, Nc=Co, flip=False, ksize=1, out_dtype=torch.bfloat16, wq=W1[3])) for _ in range(5))
    print(f"to_qkv {Ci}->{Co} @{H}: {ts[2]:6.1f} us", flush=True)
