#!/usr/bin/env python3
"""Golden training TRAJECTORY from the reference itself (PyTorch CPU, fp32): 20 Adam steps (lr 1e-4, betas .9 / .999, the values of
configs/model/ddpm.yaml) of the dim-32 / 1-2-4 UNet at 16x16, B = 4, L1 loss, on a DIFFERENT fixed (x, t, eps) per step.

    python tools/gen_golden_traj.py        # writes tests/golden/traj20.npz (inputs of every step, the loss curve, weight norms)

Build container only (imports /root/reference through tools/gen_golden.py's stubs); nothing of the reference is written out."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_golden import OUT, import_reference  # noqa: E402

STEPS, B, SIDE = 20, 4, 16
KEYS = ("final_conv.1.weight", "downs.0.0.block1.block.0.weight", "mid_attn.fn.fn.to_qkv.weight", "time_mlp.1.weight",
        "ups.1.0.res_conv.weight", "downs.1.1.block2.block.1.weight")


def main():
    R = import_reference()
    import contextlib
    import io
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        net = R.Unet(dim=32, dim_mults=(1, 2, 4), channels=3)
    gd = R.GaussianDiffusion(net, image_size=(SIDE, SIDE), timesteps=1000, loss_type="l1")
    opt = torch.optim.Adam(net.parameters(), lr=1e-4, betas=(0.9, 0.999))
    gen = torch.Generator().manual_seed(20260930)
    xs = torch.rand(STEPS, B, 3, SIDE, SIDE, generator=gen) * 2 - 1
    ts = torch.randint(0, 1000, (STEPS, B), generator=gen)
    ts[0] = torch.tensor([0, 1, 500, 999])
    ns = torch.randn(STEPS, B, 3, SIDE, SIDE, generator=gen)
    w0 = {k: v.detach().clone() for k, v in net.named_parameters()}
    losses, wnorm = [], []
    for k in range(STEPS):
        opt.zero_grad()
        loss = gd.p_losses(xs[k], ts[k], ns[k])
        loss.backward()
        opt.step()
        losses.append(float(loss))
        wnorm.append(float(torch.sqrt(sum((p.detach().double() ** 2).sum() for p in net.parameters()))))
    out = {"x": xs.numpy(), "t": ts.numpy(), "noise": ns.numpy(), "losses": np.array(losses, dtype=np.float64),
           "weight_norm": np.array(wnorm, dtype=np.float64),
           # how far the whole parameter vector moved, and a few tensors' displacement (final - initial)
           "moved_norm": np.array(float(torch.sqrt(sum(((p.detach() - w0[k_]).double() ** 2).sum() for k_, p in net.named_parameters()))))}
    params = dict(net.named_parameters())
    for k in KEYS:
        out["delta." + k] = (params[k].detach() - w0[k]).numpy()
    np.savez_compressed(os.path.join(OUT, "traj20.npz"), **out)
    print("traj20.npz:", [round(v, 5) for v in losses[:3]], "...", round(losses[-1], 5), "moved", float(out["moved_norm"]))


if __name__ == "__main__":
    main()
