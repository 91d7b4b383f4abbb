"""Out-of-bounds probe of the ops the time-MLP backward uses at a batch that is not a multiple of 32 (B = 16): every output sits between
guard zones, every input is followed by NaN padding (an op that reads past its input turns an output NaN; one that writes past its output
changes a guard)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-generation-models_amd")]
import torch
from src.ops import functional as K
dev = "cuda"
G = 4096


def guarded(shape, fill=None):
    n = 1
    for s in shape: n *= s
    buf = torch.full((n + 2 * G,), 12345.0, device=dev)
    t = buf[G:G + n].view(*shape)
    if fill is None: t.zero_()
    else: t.copy_(fill)
    return buf, t, n


def padded(t):
    """copy of t followed (and preceded) by NaNs"""
    n = t.numel()
    buf = torch.full((n + 2 * G,), float("nan"), device=dev)
    buf[G:G + n] = t.flatten()
    return buf[G:G + n].view(t.shape)


def check(name, buf, n, out, ref, tol=1e-4):
    ok_guard = bool((buf[:G] == 12345.0).all() and (buf[G + n:] == 12345.0).all())
    err = float((out - ref).abs().max() / (ref.abs().max() + 1e-30))
    fin = bool(torch.isfinite(out).all())
    print(f"{name:40s} guards intact {ok_guard}  finite {fin}  rel err {err:.2e}", flush=True)


torch.manual_seed(0)
for B in (16, 8, 24, 48):
    for (o, i) in ((32, 128), (448, 32), (128, 32)):
        dy = padded(torch.randn(B, o, device=dev)); x = padded(torch.randn(B, i, device=dev))
        buf, gw, n = guarded((o, i))
        r = K.small_gemm(True, False, dy, x, out=gw, accumulate=True, allow_split=True)
        if r is None:
            K.conv_wgrad(dy.view(B, 1, 1, o), x.view(B, 1, 1, i), gw.view(-1), kh=1, kw=1, stride=1, pad=0, gather_i=True, Ci=o, Cj=i,
                         grid_g=(1, 1), grid_d=(1, 1), mode=K.MODE_FP32)
        check(f"B{B} dW {o}x{i} ({'small_gemm' if r is not None else 'conv_wgrad'})", buf, n, gw, dy.t() @ x)
        buf, gb, n = guarded((o,))
        K.colsum(dy, gb)
        check(f"B{B} colsum {o}", buf, n, gb, dy.sum(0))
        w = padded(torch.randn(o, i, device=dev))
        dx = K.small_gemm(False, False, dy, w, allow_split=True)
        if dx is None:
            dx = K.conv_igemm(dy.view(B, 1, 1, o), w, kh=1, kw=1, stride=1, pad=0, transposed=False, w_kn=True, K=o, Nc=i, out_hw=(1, 1),
                              mode=K.MODE_FP32).view(B, i)
        print(f"B{B} dx {o}->{i}: finite {bool(torch.isfinite(dx).all())} rel err {float((dx - dy @ w).abs().max() / (dy @ w).abs().max()):.2e}", flush=True)
