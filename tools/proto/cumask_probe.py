#!/usr/bin/env python3
"""Do the HBM-bound kernels of backward need the whole chip?  Times level-0 kernels on CU-masked streams (hipExtStreamCreateWithCUMask)
and the batched 3x3 weight gradient beside a dgrad-like chain: serial on one stream against concurrent on two masked streams."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-generation-models_amd")]
import torch
from src.ops import functional as K
hip = C.CDLL("libamdhip64.so")
DEV = "cuda"
B, H, Cc = 128, 32, 128
BF = torch.bfloat16


def masked_stream(keep):
    """keep(i) -> bool for CU bit i of 256"""
    words = (C.c_uint32 * 8)()
    for i in range(256):
        if keep(i):
            words[i // 32] |= 1 << (i % 32)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


x = torch.randn(B, H, H, Cc, device=DEV).to(BF)
dy = torch.randn(B, H, H, Cc, device=DEV).to(BF)
ga = torch.ones(Cc, device=DEV); be = torch.zeros(Cc, device=DEV); tb = torch.randn(B, Cc, device=DEV)
dg, db, dbias, dtb = torch.zeros(Cc, device=DEV), torch.zeros(Cc, device=DEV), torch.zeros(Cc, device=DEV), torch.zeros(B, Cc, device=DEV)
yy, st = K.gn_mish_fwd(x, ga, be, temb=tb, out_dtype=BF)
w = torch.randn(3, 3, Cc, Cc, device=DEV) * 0.05
table, nent, tiles = K.pack_table([(0, 9, Cc, Cc)], DEV)
WQ = [torch.zeros(w.numel(), device=DEV, dtype=BF) for _ in range(4)]
K.pack_weights_bf16(table, nent, tiles, w.reshape(-1), *WQ)
Y = torch.empty(B, H, H, Cc, device=DEV, dtype=BF)
qkv = torch.randn(B, H, H, 384, device=DEV).to(BF)
o, ctx, ks = K.linattn_fwd(qkv)
do = torch.randn(B, H, H, 128, device=DEV).to(BF)
Q8 = []
for (h, ci, co) in [(32, 128, 128)] * 4 + [(16, 256, 256)] * 2 + [(16, 512, 128), (16, 256, 256)]:
    Q8.append((torch.randn(B, h, h, ci, device=DEV).bfloat16(), torch.randn(B, h, h, co, device=DEV).bfloat16(),
               torch.zeros(9 * ci * co, device=DEV), h, ci, co))


def gn_bwd():
    K.gn_mish_bwd(x, st, ga, be, dy, dgamma=dg, dbeta=db, dtemb=dtb, dbias=dbias, out_dtype=BF)


def gn_fwd():
    K.gn_mish_fwd(x, ga, be, temb=tb, out_dtype=BF)


def conv():
    K.conv3x3_bf16w(x, WQ[1], K=Cc, Nc=Cc, flip=False, out=Y, wq=WQ[3])


def attn_bwd():
    K.linattn_bwd(qkv, ctx, ks, do)


def wgradq():
    q = K.WgradQueue(group=8)
    for (xx, dd, ww, h, ci, co) in Q8:
        q.push(xx, dd, ww, Ci=ci, Cj=co, hw=(h, h), mode=1)
    q.flush()


def chain():
    for _ in range(4):
        conv(); gn_bwd(); conv(); gn_bwd(); attn_bwd()


def timeit(fn, stream, n=20):
    with torch.cuda.stream(stream):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(n):
            fn()
        e1.record(stream)
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


try:
    conv()
except Exception as e:                                            # noqa: BLE001
    print("conv entry:", e)
    sys.exit(1)
full = torch.cuda.current_stream()
masks = {"256": None, "224": lambda i: (i // 8) % 8 != 7, "192": lambda i: (i // 8) % 4 != 3, "128": lambda i: (i // 8) % 2 == 0}
streams = {k: (full if m is None else masked_stream(m)) for k, m in masks.items()}
for name, fn in (("conv_pw", conv), ("gn_bwd", gn_bwd), ("gn_fwd", gn_fwd), ("linattn_bwd", attn_bwd), ("wgradq(8)", wgradq)):
    print(f"{name:12s} " + "  ".join(f"{k} CUs {timeit(fn, s):7.1f} us" for k, s in streams.items()), flush=True)

# serial: chain + one batched weight gradient on the full chip;  concurrent: the weight gradient on a side stream
t_chain, t_w = timeit(chain, full, 10), timeit(wgradq, full, 10)
print(f"chain {t_chain:.1f} us, wgradq {t_w:.1f} us, serial sum {t_chain + t_w:.1f} us")
for label, smain, sside in (("both unmasked", full, torch.cuda.Stream()),
                            ("192 | 64", streams["192"], masked_stream(lambda i: (i // 8) % 4 == 3)),
                            ("224 | 32", streams["224"], masked_stream(lambda i: (i // 8) % 8 == 7)),
                            ("256 | 64", full, masked_stream(lambda i: (i // 8) % 4 == 3)),
                            ("256 | 32", full, masked_stream(lambda i: (i // 8) % 8 == 7))):
    ts = []
    for rep in range(4):
        torch.cuda.synchronize()
        e0, e1, ej = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event()
        with torch.cuda.stream(smain):
            e0.record(smain)
            sside.wait_event(e0)
            with torch.cuda.stream(sside):
                wgradq()
                ej.record(sside)
            chain()
            smain.wait_event(ej)
            e1.record(smain)
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    print(f"concurrent [{label}]: {min(ts[1:]):.1f} us")
