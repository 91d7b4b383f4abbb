#!/usr/bin/env python3
"""T=1000 sampler vectors from the REFERENCE (north_star: parity of "sampled images"; reference src/models/ddpm.py:399-415).

Runs only in the build container (imports /root/reference through tools/gen_golden.import_reference).  For two UNets --
tiny (dim 8, mults 1-2, 3x8x8; the weights of tests/golden/tiny_unet.npz) and mid (dim 32, mults 1-2-4, 3x16x16, seeded default
init torch.manual_seed(0)) -- it runs the reference's own `GaussianDiffusion.sample(2)` with T=1000 under
`torch.manual_seed(SEED)` and stores the final images plus the images after 250 / 500 / 750 reverse steps.  The host noise
tape (1001 draws of randn(2,3,H,W) from torch's CPU mt19937 generator: x_T first, then one per step, ddpm.py:268-273,404-408)
is NOT stored -- 1.5 MB of noise per case; the test re-draws it from the same seed and checks the SHA-256 stored here before
using it, so a torch whose CPU generator drew differently fails loudly instead of comparing against the wrong tape.

    python tools/gen_golden_t1000.py      # writes tests/golden/t1000_sampler.npz
"""
import hashlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_golden import OUT, import_reference, np_, quiet   # noqa: E402

SEED = 4242
T = 1000
MARKS = (250, 500, 750)          # reverse steps done when the intermediate image is stored


def run_case(R, net, hw):
    gd = R.GaussianDiffusion(net, image_size=(hw, hw), timesteps=T)
    shape = (2, 3, hw, hw)
    torch.manual_seed(SEED)
    h = hashlib.sha256()
    for _ in range(T + 1):
        h.update(torch.randn(shape).numpy().tobytes())
    inter = {}
    orig = gd.p_sample
    count = [0]

    def spy(x, t, *a, **k):
        out = orig(x, t, *a, **k)
        count[0] += 1
        if count[0] in MARKS:
            inter[count[0]] = np_(out)
        return out
    gd.p_sample = spy
    R.tqdm = lambda it, **k: it
    torch.manual_seed(SEED)
    with torch.no_grad():
        s = gd.sample(2)
    assert count[0] == T
    return np_(s), inter, h.hexdigest()


def main():
    torch.set_num_threads(8)
    R = import_reference()
    out = {"seed": np.int64(SEED), "T": np.int64(T), "marks": np.array(MARKS)}
    # tiny: same weights as tiny_unet.npz (seed 0 default init of the reference)
    torch.manual_seed(0)
    net = quiet(R.Unet, dim=8, dim_mults=(1, 2), channels=3).eval()
    g = np.load(os.path.join(OUT, "tiny_unet.npz"))
    for k, v in net.state_dict().items():
        assert np.array_equal(np_(v), g["w." + k]), k
    s, inter, sha = run_case(R, net, 8)
    out["tiny.sample"] = s; out["tiny.tape_sha256"] = np.array(sha)
    for m, v in inter.items():
        out[f"tiny.after{m}"] = v
    print("tiny: sum", float(s.sum()), "absmax", float(np.abs(s).max()), sha[:16])
    # mid: seeded default init, same as mid_unet.npz's network
    torch.manual_seed(0)
    net = quiet(R.Unet, dim=32, dim_mults=(1, 2, 4), channels=3).eval()
    s, inter, sha = run_case(R, net, 16)
    out["mid.sample"] = s; out["mid.tape_sha256"] = np.array(sha)
    for m, v in inter.items():
        out[f"mid.after{m}"] = v
    print("mid: sum", float(s.sum()), "absmax", float(np.abs(s).max()), sha[:16])
    np.savez_compressed(os.path.join(OUT, "t1000_sampler.npz"), **out)
    print("wrote t1000_sampler.npz", os.path.getsize(os.path.join(OUT, "t1000_sampler.npz")))


if __name__ == "__main__":
    main()
