"""Shared helpers for the GPU parity tests."""
import numpy as np
import torch

DEV = "cuda"


def to_nhwc_gpu(x_nchw: torch.Tensor, pad4: bool = True) -> torch.Tensor:
    """CPU NCHW -> GPU NHWC activation with a 4-aligned pixel stride (plain torch copy: test plumbing)."""
    n, c, h, w = x_nchw.shape
    ld = (c + 3) // 4 * 4 if pad4 else c
    buf = torch.zeros((n, h, w, ld), dtype=torch.float32)
    buf[..., :c] = x_nchw.permute(0, 2, 3, 1).float()
    return buf.to(DEV)[..., :c]


def from_nhwc(x: torch.Tensor) -> torch.Tensor:
    return x.detach().cpu().permute(0, 3, 1, 2).contiguous()


def conv_w_storage(w_logical: torch.Tensor, transposed=False) -> torch.Tensor:
    """PyTorch [Cout,Cin,kh,kw] (or ConvT [Cin,Cout,kh,kw]) -> tap-major [kh,kw,Cin,Cout] on the GPU."""
    if transposed:
        return w_logical.permute(2, 3, 0, 1).contiguous().float().to(DEV)
    return w_logical.permute(2, 3, 1, 0).contiguous().float().to(DEV)


def w_from_storage(ws: torch.Tensor, transposed=False) -> torch.Tensor:
    ws = ws.detach().cpu()
    return ws.permute(2, 3, 0, 1).contiguous() if transposed else ws.permute(3, 2, 0, 1).contiguous()


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def max_err(a, b) -> float:
    return float((a.double().cpu() - b.double().cpu()).abs().max())


TOL = {0: 2e-5, 1: 2e-2}     # rel-L2 tolerance per mode: exact-fp32 MFMA / bf16 MFMA (fp32 accumulate)
