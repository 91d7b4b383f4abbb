#!/usr/bin/env python3
"""One conv_pw layer launched N times and nothing else (for rocprofv3 PC sampling / counters): H Ci Co [tile] [n] [out16]."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-generation-models_amd")]
import torch
from src.ops import functional as K
H, Ci, Co = (int(v) for v in sys.argv[1:4])
tile = int(sys.argv[4]) if len(sys.argv) > 4 else 0
n = int(sys.argv[5]) if len(sys.argv) > 5 else 100
out16 = (sys.argv[6] != "0") if len(sys.argv) > 6 else True
B = int(os.environ.get("B", 128))
lib = K.load_library(); K.PW_MIN_TILES = 0
lib.mi_debug_conv_pw_tile(tile)
x = torch.randn(B, H, H, Ci, device="cuda").bfloat16()
w = torch.randn(3, 3, Ci, Co, device="cuda") * 0.05
table, nent, tiles = K.pack_table([(0, 9, Ci, Co)], "cuda")
wd, wf, wdq, wfq = (torch.zeros(w.numel(), device="cuda", dtype=torch.bfloat16) for _ in range(4))
K.pack_weights_bf16(table, nent, tiles, w.reshape(-1), wd, wf, wdq, wfq)
y = torch.empty(B, H, H, Co, device="cuda", dtype=torch.bfloat16 if out16 else torch.float32)
for _ in range(n):
    K.conv3x3_bf16w(x, wf, K=Ci, Nc=Co, flip=False, out=y, wq=wfq)
torch.cuda.synchronize()
print("done")
