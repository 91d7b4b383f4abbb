#!/usr/bin/env python3
"""Micro-benchmark of the 1x1 weight-gradient launches (mi_conv1x1_wgrad_tr_batch) on the cfg-2 layers at B = 128: every layer alone and
the launches backward actually issues (the to_out gradient goes out alone, before the LayerNorm backward overwrites its operand).
MI_DDPM_LIB selects the library, so two builds can be compared on one box:
    MI_DDPM_LIB=.../libmi_ddpm_r03.so python tools/bench_w1.py; python tools/bench_w1.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-generation-models_amd")]
import torch
from src.ops import functional as K

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
DEV = "cuda"
# (H, Ci, Co, fp32 dY, two-source split)
LAYERS = {
    "to_out 128->128 @16": (16, 128, 128, True, None), "to_qkv 128->384 @16": (16, 128, 384, False, None),
    "res 512->128 @16": (16, 512, 128, True, 256), "to_out 128->256 @8": (8, 128, 256, True, None),
    "to_qkv 256->384 @8": (8, 256, 384, False, None), "res 1024->256 @8": (8, 1024, 256, True, 512),
    "to_out 128->512 @8": (8, 128, 512, True, None), "to_qkv 512->384 @8": (8, 512, 384, False, None),
    "res 256->512 @8": (8, 256, 512, True, None), "to_out 128->256 @16": (16, 128, 256, True, None),
    "to_qkv 256->384 @16": (16, 256, 384, False, None), "res 128->256 @16": (16, 128, 256, True, None),
    "to_out 128->128 @32": (32, 128, 128, True, None), "to_qkv 128->384 @32": (32, 128, 384, False, None),
}
# the launches of one backward pass (cfg 2): groups of layer names
LAUNCHES = [["to_out 128->128 @16"], ["to_qkv 128->384 @16", "res 512->128 @16", "to_out 128->256 @8"],
            ["to_qkv 256->384 @8", "res 1024->256 @8", "to_out 128->512 @8"], ["to_qkv 512->384 @8", "to_out 128->512 @8"],
            ["to_qkv 512->384 @8", "res 256->512 @8", "to_out 128->256 @16"], ["to_qkv 256->384 @16", "res 128->256 @16", "to_out 128->128 @32"],
            ["to_qkv 128->384 @32"]]


def make(name):
    H, Ci, Co, q32, split = LAYERS[name]
    x = torch.randn(B, H, H, Ci, device=DEV).bfloat16()
    P, P2 = (x[..., :split], x[..., split:]) if split else (x, None)
    Q = torch.randn(B, H, H, Co, device=DEV)
    Q = Q if q32 else Q.bfloat16()
    dW = torch.zeros(Ci * Co, device=DEV)
    db = torch.zeros(Co, device=DEV) if q32 else None
    nbytes = B * H * H * (Ci * 2 + Co * (4 if q32 else 2))
    return dict(P=P, P2=P2, Q=Q, dW=dW, db=db, Ci=Ci, Co=Co, H=H, nbytes=nbytes)


def run(items):
    q = K.WgradQueue(group=8)
    for it in items:
        q.push1x1(it["P"], it["Q"], it["dW"], Ci=it["Ci"], Cj=it["Co"], hw=(it["H"], it["H"]), mode=1, P2=it["P2"], dbias=it["db"])
    q.flush()


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


made = {k: make(k) for k in LAYERS}
print(f"lib = {os.environ.get('MI_DDPM_LIB', 'default')}")
for k, it in made.items():
    t = timeit(lambda: run([it]))
    print(f"  {k:22s} {t*1e6:7.1f} us  {it['nbytes']/t/1e9:7.0f} GB/s algorithmic", flush=True)
tot = 0.0
for grp in LAUNCHES:
    items = [made[k] for k in grp]
    t = timeit(lambda: run(items))
    tot += t
    nb = sum(it["nbytes"] for it in items)
    print(f"  launch {'+'.join(grp)}: {t*1e6:7.1f} us  {nb/t/1e9:7.0f} GB/s", flush=True)
print(f"  sum of the step's launches (kernel + reduce): {tot*1e3:.3f} ms")
