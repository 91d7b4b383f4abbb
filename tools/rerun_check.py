#!/usr/bin/env python3
"""Forward determinism of the cfg-3 UNet at B = 32 in bf16 mode under kernel-pick variants (debugging aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-generation-models_amd")]
import torch
from src.ops import functional as K
from src.models.ddpm import Unet
torch.manual_seed(0)
net = Unet(dim=64, dim_mults=(1, 2, 4, 8), channels=3); net.compute_mode = "bf16"; net = net.to("cuda").eval()
g = torch.Generator().manual_seed(16)
x = (torch.rand(32, 3, 64, 64, generator=g) * 2 - 1).cuda(); t = torch.randint(0, 1000, (32,), generator=g).cuda()
rel = lambda a, b: float((a - b).norm() / b.norm())
def run(tag):
    with torch.no_grad():
        ys = [net(x, t) for _ in range(4)]
    print(tag, [f"{rel(y, ys[0]):.2e}" for y in ys[1:]], flush=True)
    return ys[0]
y0 = run("default        ")
orig = K.conv3x3_pw_gn_mish_picked
for hh in (32, 16, 8):
    K.conv3x3_pw_gn_mish_picked = lambda N, H, W, Kc, Nc, hh=hh: H == hh and orig(N, H, W, Kc, Nc)
    run(f"fused only H={hh:2d}")
K.conv3x3_pw_gn_mish_picked = orig
net.fuse_gn_conv = "2"; run("fuse=2 (forced) ")
net.fuse_gn_conv = "1"; run("fuse=1 (stats pass)")
net.fuse_gn_conv = "0"; run("fuse off       ")
