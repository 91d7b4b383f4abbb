#!/usr/bin/env python3
"""Run ONE kernel configuration a few times (for rocprofv3 counter passes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-generation-models_amd")]
import torch
from src.ops import functional as K
which, B, H, Ci, Co = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
DT = torch.bfloat16 if len(sys.argv) > 6 and sys.argv[6] == "bf16" else torch.float32     # activation storage
DEV = "cuda"
H, Ci, Co = max(H, 8), max(Ci, 64), max(Co, 64)
x = torch.randn(B, H, H, Ci, device=DEV).to(DT); w = torch.randn(3, 3, Ci, Co, device=DEV) * 0.05
dy = torch.randn(B, H, H, Co, device=DEV).to(DT)
wd = w.to(torch.bfloat16).reshape(-1); wf = w.permute(0, 1, 3, 2).contiguous().to(torch.bfloat16).reshape(-1)
y = torch.empty(B, H, H, Co, device=DEV, dtype=DT); dW = torch.zeros(9 * Ci * Co, device=DEV)
for _ in range(5):
    if which == "wgrad":
        K.conv_wgrad(x, dy, dW, kh=3, kw=3, stride=1, pad=1, gather_i=True, Ci=Ci, Cj=Co, grid_g=(H, H), grid_d=(H, H), mode=1)
    elif which == "wgradq":
        # the eight Block-conv weight gradients of a backward pass's first group in ONE launch (K.WgradQueue), cfg-2 shapes
        if "Q8" not in globals():
            Q8 = []
            for (h, ci, co) in [(32, 128, 128)] * 4 + [(16, 256, 256)] * 2 + [(16, 512, 128), (16, 256, 256)]:
                Q8.append((torch.randn(B, h, h, ci, device=DEV).bfloat16(), torch.randn(B, h, h, co, device=DEV).bfloat16(),
                           torch.zeros(9 * ci * co, device=DEV), h, ci, co))
        q = K.WgradQueue(group=8)
        for (xx, dd, ww, h, ci, co) in Q8:
            q.push(xx, dd, ww, Ci=ci, Cj=co, hw=(h, h), mode=1)
        q.flush()
    elif which == "fused":
        if "COEF" not in globals():
            gam = torch.ones(Ci, device=DEV); bet = torch.zeros(Ci, device=DEV); tb = torch.randn(B, Ci, device=DEV) * 0.1
            _, COEF = K.gn_stats_coef(x, gam, bet, temb=tb)
        K.conv3x3_gn_mish(x, COEF, wf, K=Ci, Nc=Co, bias=None)
    elif which == "fusedpw":     # the named fused kernel on the private-weight-stream structure (coefficient tensor given)
        if "COEF" not in globals():
            gam = torch.ones(Ci, device=DEV); bet = torch.zeros(Ci, device=DEV); tb = torch.randn(B, Ci, device=DEV) * 0.1
            _, COEF = K.gn_stats_coef(x, gam, bet, temb=tb)
            table, nent, tiles = K.pack_table([(0, 9, Ci, Co)], DEV)
            WQ = [torch.zeros(w.numel(), device=DEV, dtype=torch.bfloat16) for _ in range(4)]
            K.pack_weights_bf16(table, nent, tiles, w.reshape(-1), *WQ)
        K.conv3x3_gn_mish(x, COEF, WQ[1], K=Ci, Nc=Co, bias=None, wq=WQ[3])
    elif which == "gn":          # GroupNorm+Mish of a Block (bf16 in / out) forward + backward on [B,H,H,Ci]
        if "GN" not in globals():
            ga = torch.ones(Ci, device=DEV); be = torch.zeros(Ci, device=DEV); tb = torch.randn(B, Ci, device=DEV)
            GN = (ga, be, tb, torch.zeros(Ci, device=DEV), torch.zeros(Ci, device=DEV), torch.zeros(Ci, device=DEV), torch.zeros(B, Ci, device=DEV))
        ga, be, tb, dg, db, dbias, dtb = GN
        yy, st = K.gn_mish_fwd(x, ga, be, temb=tb, out_dtype=DT)
        K.gn_mish_bwd(x, st, ga, be, yy, dgamma=dg, dbeta=db, dtemb=dtb, dbias=dbias, out_dtype=DT)
    elif which == "ln":          # channel LayerNorm of PreNorm (fp32 stream in, bf16 out) forward + backward
        if "LN" not in globals():
            LN = (torch.randn(B, H, H, Ci, device=DEV), torch.ones(Ci, device=DEV), torch.zeros(Ci, device=DEV),
                  torch.zeros(B, H, H, Ci, device=DEV), torch.zeros(Ci, device=DEV), torch.zeros(Ci, device=DEV))
        xs, g1, b1, dxs, dg1, db1 = LN
        yy = K.chan_layernorm_fwd(xs, g1, b1, out_dtype=torch.bfloat16)
        q4 = K.WgradQueue()                      # as Unet's backward runs it: dg / db as partial rows + the batched row sum
        K.chan_layernorm_bwd(xs, g1, yy, dxs, True, dg1, db1, defer=q4)
        q4.flush()
    elif which == "attn":        # LinearAttention core on bf16 qkv [B,H,H,384]
        if "QKV" not in globals():
            QKV = torch.randn(B, H, H, 384, device=DEV).bfloat16()
        ao, ctx, kst = K.linattn_fwd(QKV, 4)
        K.linattn_bwd(QKV, ctx, kst, ao, 4)
    elif which == "c1x1":        # to_qkv (Ci -> Co, bf16 in / out) forward through the tile kernel
        if "W1" not in globals():
            w1 = torch.randn(1, 1, Ci, Co, device=DEV) * 0.05
            table, nent, tiles = K.pack_table([(0, 1, Ci, Co)], DEV)
            W1 = [torch.zeros(w1.numel(), device=DEV, dtype=torch.bfloat16) for _ in range(4)]
            K.pack_weights_bf16(table, nent, tiles, w1.reshape(-1), *W1)
        K.conv3x3_bf16w(x, W1[1], K=Ci, Nc=Co, flip=False, ksize=1, out_dtype=DT, wq=W1[3])
    elif which == "halo":        # the register-staged halo kernel (not the per-shape pick)
        K.CONV_AUTO = False
        K.conv3x3_bf16w(x, wf, K=Ci, Nc=Co, flip=False, out=y)
    elif which == "pw":          # the private-weight-stream kernel (conv_pw.hip), the default of the bf16-stored layers with >= 200 tiles
        if "WQ" not in globals():
            table, nent, tiles = K.pack_table([(0, 9, Ci, Co)], DEV)
            WQ = [torch.zeros(w.numel(), device=DEV, dtype=torch.bfloat16) for _ in range(4)]
            K.pack_weights_bf16(table, nent, tiles, w.reshape(-1), *WQ)
        K.conv3x3_bf16w(x, WQ[1], K=Ci, Nc=Co, flip=False, out=y, wq=WQ[3])
    elif which in ("gtdown", "gtup"):     # the tap-gather kernel: Downsample (3x3 / stride 2) on [B,H,H,Ci], Upsample (4x4 / stride 2 transposed)
        k = 3 if which == "gtdown" else 4
        if "WG" not in globals():
            wg = torch.randn(k, k, Ci, Co, device=DEV) * 0.05
            table, nent, tiles = K.pack_table([(0, k * k, Ci, Co)], DEV)
            WG = [torch.zeros(wg.numel(), device=DEV, dtype=torch.bfloat16) for _ in range(4)]
            K.pack_weights_bf16(table, nent, tiles, wg.reshape(-1), *WG)
            YG = torch.empty(B, H // 2 if which == "gtdown" else 2 * H, H // 2 if which == "gtdown" else 2 * H, Co, device=DEV)
        K.conv_gt(x.bfloat16() if x.dtype != torch.bfloat16 else x, WG[3], kh=k, kw=k, stride=2, pad=1, transposed=which == "gtup", K=Ci, Nc=Co,
                  out_hw=(YG.shape[1], YG.shape[2]), out=YG)
    elif which == "igemm":
        K.conv_igemm(x, w, kh=3, kw=3, stride=1, pad=1, transposed=False, w_kn=True, K=Ci, Nc=Co, out_hw=(H, H), mode=1, out=y)
torch.cuda.synchronize()
