#!/usr/bin/env python3
"""Why the bf16-mode gradient of `mid_attn.fn.fn.to_out.bias` is the worst per-tensor gradient (9 %): a bias gradient is a column sum of
the output gradient dY, and a sum of mixed-sign summands amplifies their relative error by  A_c = sum |dY_c| / |sum dY_c|  per channel c.
This measures A_c (fp64, the CPU oracle, tests/golden/mid_unet.npz's inputs on the seeded mid UNet of tests/test_unet_gpu.py) for every
conv / attention bias of the network by hooking the conv outputs' gradients.  CPU only (uses oracle/: test infrastructure).
    python tools/bias_grad_amplification.py            -> a table + profiles/r04_bias_grad_amplification.json"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-generation-models_amd")]
from oracle import ddpm_oracle as O  # noqa: E402

g = np.load(os.path.join(ROOT, "tests", "golden", "mid_unet.npz"))
torch.manual_seed(0)
p = O.init_unet_params(32, (1, 2, 4), 3)
p = {k: v.double().requires_grad_(True) for k, v in p.items()}
name_of = {id(v): k for k, v in p.items()}
grads = {}
real_conv2d = F.conv2d


def hooked(x, w, b=None, *a, **kw):
    y = real_conv2d(x, w, b, *a, **kw)
    if b is not None and id(b) in name_of:
        key = name_of[id(b)]
        y.register_hook(lambda gr, key=key: grads.__setitem__(key, gr.detach()))
    return y


O.F.conv2d = hooked
tab = {k: v.double() if v.is_floating_point() else v for k, v in O.schedule_tables(1000).items()}
x, t, noise = (torch.from_numpy(g[k]) for k in ("x", "t", "noise"))
loss, _ = O.p_losses(p, tab, x.double(), t, noise.double())
loss.backward()
rows = {}
for key, dy in grads.items():
    s_abs = dy.abs().sum((0, 2, 3)); s = dy.sum((0, 2, 3))
    amp = (s_abs / s.abs().clamp_min(1e-300))
    assert torch.allclose(s, p[key].grad, rtol=1e-9, atol=1e-12)
    # amplification of the whole tensor's relative L2 error when every summand carries an independent relative error eps:
    # ||delta g|| / ||g|| ~ eps * sqrt(sum_c sum dY^2) / ||g||
    l2 = float(torch.sqrt((dy * dy).sum()) / s.norm())
    rows[key] = {"median_Ac": float(amp.median()), "max_Ac": float(amp.max()), "tensor_l2_amplification": l2, "channels": int(s.numel())}
order = sorted(rows, key=lambda k: -rows[k]["tensor_l2_amplification"])
print(f"{'bias':44s} {'median A_c':>11s} {'max A_c':>10s} {'sqrt(sum dY^2)/||g||':>22s}")
for k in order[:12]:
    r = rows[k]
    print(f"{k:44s} {r['median_Ac']:11.1f} {r['max_Ac']:10.1f} {r['tensor_l2_amplification']:22.2f}")
out = os.path.join(ROOT, "profiles", "r04_bias_grad_amplification.json")
json.dump({"note": __doc__.strip().splitlines()[0], "rows": {k: rows[k] for k in order}}, open(out, "w"), indent=1)
print("->", out)
