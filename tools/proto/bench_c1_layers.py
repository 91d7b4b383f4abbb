#!/usr/bin/env python3
"""The 1x1 convs of the cfg-2 step through conv1x1_pw at B = 128: to_qkv (bf16 out), to_out (fp32 out + residual + bf16 copy), res_conv (fp32 out)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-generation-models_amd")]
import torch
from src.ops import functional as K
B = 128


def timed(run, n=30):
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def pack(Ci, Co):
    w1 = torch.randn(1, 1, Ci, Co, device="cuda") * 0.05
    table, nent, tiles = K.pack_table([(0, 1, Ci, Co)], "cuda")
    W = [torch.zeros(w1.numel(), device="cuda", dtype=torch.bfloat16) for _ in range(4)]
    K.pack_weights_bf16(table, nent, tiles, w1.reshape(-1), *W)
    return W


tot = 0.0
for (H, Ci, Co, kind) in [(32, 128, 384, "qkv"), (16, 256, 384, "qkv"), (8, 512, 384, "qkv"), (32, 128, 128, "out"), (16, 128, 256, "out"), (8, 128, 512, "out"),
                          (32, 384, 128, "dqkv"), (16, 384, 256, "dqkv"), (8, 384, 512, "dqkv"), (16, 128, 256, "res"), (8, 256, 512, "res")]:
    x = torch.randn(B, H, H, Ci, device="cuda").bfloat16()
    W = pack(Ci, Co)
    if kind in ("qkv", "dqkv"):
        run = lambda: K.conv3x3_bf16w(x, W[1], K=Ci, Nc=Co, flip=False, ksize=1, out_dtype=torch.bfloat16, wq=W[3])
        nbytes = x.numel() * 2 + B * H * H * Co * 2
    elif kind == "out":
        res = torch.randn(B, H, H, Co, device="cuda")
        run = lambda: K.conv3x3_bf16w(x, W[1], K=Ci, Nc=Co, flip=False, ksize=1, residual=res, want16=True, wq=W[3])
        nbytes = x.numel() * 2 + B * H * H * Co * (4 + 4 + 2)
    else:
        run = lambda: K.conv3x3_bf16w(x, W[1], K=Ci, Nc=Co, flip=False, ksize=1, wq=W[3])
        nbytes = x.numel() * 2 + B * H * H * Co * 4
    ts = sorted(timed(run) for _ in range(5))
    tot += ts[2]
    print(f"{kind:5s} {Ci:4d}->{Co:4d} @{H:2d}: {ts[2]:6.1f} us  {nbytes / ts[2] / 1e6:5.2f} TB/s", flush=True)
print(f"sum {tot:.1f} us")
