"""GroupNorm+Mish forward / backward at constant bytes (16.8 M bf16 elements) and constant slice size, varying only the
contiguous bytes a group owns per pixel (32 / 64 / 128 / 256 B)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-generation-models_amd")]
import torch
from src.ops import functional as K


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


dt = torch.bfloat16
for rep in range(2):
    for B, H, C in [(128, 32, 128), (256, 32, 64), (128, 16, 512), (256, 16, 256), (512, 16, 128), (512, 8, 512)]:
        x = torch.randn(B, H, H, C, device="cuda").to(dt)
        ga = torch.ones(C, device="cuda"); be = torch.zeros(C, device="cuda")
        tb = torch.randn(B, C, device="cuda")
        y, st = K.gn_mish_fwd(x, ga, be, temb=tb, out_dtype=dt)
        tf = timeit(lambda: K.gn_mish_fwd(x, ga, be, temb=tb, out_dtype=dt))
        dg, db, dbias = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        dtb = torch.zeros(B, C, device="cuda")
        tbw = timeit(lambda: K.gn_mish_bwd(x, st, ga, be, y, dgamma=dg, dbeta=db, dtemb=dtb, dbias=dbias, out_dtype=dt))
        el = x.numel()
        print(f"[{B},{H},{H},{C}] seg {C // 8 * 2:4d} B slice {H * H * C // 8:6d} el, {el / 1e6:.1f} M el: fwd {tf:6.1f} us {el * 4 / tf / 1e6:6.2f} TB/s | bwd {tbw:6.1f} us {el * 6 / tbw / 1e6:6.2f} TB/s",
              flush=True)
