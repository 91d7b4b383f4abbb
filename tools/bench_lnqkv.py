#!/usr/bin/env python3
"""LayerNorm + to_qkv: the fused inference kernel (mi_ln_conv1x1_pw) against the two launches it replaces, device time from a replayed
graph of REP calls.  usage: python tools/bench_lnqkv.py [B]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-generation-models_amd"), os.path.join(ROOT, "tests")]
import math, torch
from src.ops import functional as K
DEV = "cuda"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
REP = 20


def timed(fn):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(g, stream=s):
            for _ in range(REP):
                fn()
    for _ in range(3):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(7):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / REP)
    return best


from test_kernels_gpu import _pack, conv_w_storage
for H, C in ((32, 128), (16, 256), (8, 512)):
    x = torch.randn(B, H, H, C, device=DEV)
    gg, bb = torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV)
    w = torch.randn(384, C, 1, 1) / math.sqrt(C)
    flat, wd, wf, offs, wdq, wfq = _pack(K, [conv_w_storage(w.double())], frag=True)
    t_f = timed(lambda: K.ln_conv1x1(x, gg, bb, wfq, Nc=384))
    t_ln = timed(lambda: K.chan_layernorm_fwd(x, gg, bb, out_dtype=torch.bfloat16))
    ln = K.chan_layernorm_fwd(x, gg, bb, out_dtype=torch.bfloat16)
    t_c = timed(lambda: K.conv3x3_bf16w(ln, wf, K=C, Nc=384, flip=False, ksize=1, out_dtype=torch.bfloat16, wq=wfq))
    mb = (B * H * H * (C * 4 + 384 * 2)) / 1e6
    print(f"B={B} {H}x{H} C={C}: fused {t_f:.1f} us ({mb / t_f / 1e3 * 1e3:.0f} GB/s)   LayerNorm {t_ln:.1f} + to_qkv {t_c:.1f} = {t_ln + t_c:.1f} us")
