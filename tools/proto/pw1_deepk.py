#!/usr/bin/env python3
"""Upper bound for a tap-gather GEMM on the conv1x1_pw structure: plain 1x1 convs with the K of a 9- / 16-tap layer."""
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
for (B, H, Ci, Co, what) in [(128, 16, 1152, 128, "down 128 32->16"), (128, 8, 2304, 256, "down 256 16->8"), (128, 16, 512, 128, "convT class 128 (one of 4)"),
                             (128, 8, 1024, 256, "convT class 256"), (32, 64, 640, 64, "cfg3 64->64 @64 (K=576 padded)")]:
    x = torch.randn(B, H, H, Ci, device="cuda").bfloat16()
    w1 = torch.randn(1, 1, Ci, Co, device="cuda") * 0.05
    table, nent, tiles = K.pack_table([(0, 1, Ci, Co)], "cuda")
    W1 = [torch.zeros(w1.numel(), device="cuda", dtype=torch.bfloat16) for _ in range(4)]
    K.pack_weights_bf16(table, nent, tiles, w1.reshape(-1), *W1)
    for dt in (torch.bfloat16, torch.float32):
        t = timed(lambda: K.conv3x3_bf16w(x, W1[1], K=Ci, Nc=Co, flip=False, ksize=1, out_dtype=dt, wq=W1[3]))
        fl = 2.0 * B * H * H * Ci * Co
        print(f"{what:34s} M={B*H*H} K={Ci} Nc={Co} out {str(dt)[6:]:8s}: {t:6.1f} us {fl/t/1e6:6.0f} TF", flush=True)
