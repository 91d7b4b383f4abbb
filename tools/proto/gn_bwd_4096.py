import os, sys
sys.path[:0] = ["/root/repo", "/root/repo/image-generation-models_amd"]
import torch
from src.ops import functional as K
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B, H, C = 32, 64, 64
dt = torch.bfloat16
x = torch.randn(B, H, H, C, device="cuda").to(dt)
ga = torch.ones(C, device="cuda"); be = torch.zeros(C, device="cuda"); tb = torch.randn(B, C, device="cuda")
y, st = K.gn_mish_fwd(x, ga, be, temb=tb, out_dtype=dt)
dg, db, dbias = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
dtb = torch.zeros(B, C, device="cuda")
d32 = torch.randn(B, H, H, C, device="cuda")
print("bwd dout bf16: %.1f us   dout fp32: %.1f us   fwd %.1f us" % (
    timeit(lambda: K.gn_mish_bwd(x, st, ga, be, y, dgamma=dg, dbeta=db, dtemb=dtb, dbias=dbias, out_dtype=dt)),
    timeit(lambda: K.gn_mish_bwd(x, st, ga, be, d32, dgamma=dg, dbeta=db, dtemb=dtb, dbias=dbias, out_dtype=dt)),
    timeit(lambda: K.gn_mish_fwd(x, ga, be, temb=tb, out_dtype=dt))))
