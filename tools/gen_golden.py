#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE itself (PyTorch CPU, fp32).

Runs only in the build container: it imports /root/reference/src/models/ddpm.py at run
time (with import stubs for the wheels the image lacks: torchvision, pytorch_lightning,
omegaconf -- SURVEY.md section 8(c)) and writes plain numpy arrays + JSON under tests/golden/.
No reference source, bytecode or pickled module is written anywhere.

    python tools/gen_golden.py            # rewrites tests/golden/*.npz, pins.json
"""
import hashlib
import json
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def import_reference():
    class _LM(torch.nn.Module):
        def save_hyperparameters(self, *a, **k):
            pass

        def log(self, *a, **k):
            pass

    pl = _stub("pytorch_lightning", LightningModule=_LM, LightningDataModule=object, Callback=object,
               Trainer=object, seed_everything=torch.manual_seed)
    pl.loggers = _stub("pytorch_lightning.loggers", Logger=object)
    pl.utilities = _stub("pytorch_lightning.utilities", rank_zero_only=lambda f: f)
    _stub("torchvision")
    _stub("omegaconf", DictConfig=dict, OmegaConf=object)
    sys.path.insert(0, REF)
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        from src.models import ddpm
    return ddpm


def sha_state(sd):
    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(k.encode())
        h.update(v.detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


def quiet(fn, *a, **k):
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def np_(t):
    return t.detach().cpu().numpy().copy()


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    R = import_reference()
    pins = {"torch": torch.__version__, "init_sha256": {}, "param_count": {}}

    # ---- init-order pins (no payload) ------------------------------------------------
    for name, (dim, mults, ch) in {
        "cfg2_dim128_m124_c3": (128, (1, 2, 4), 3),
        "cfg3_dim64_m1248_c3": (64, (1, 2, 4, 8), 3),
        "mnist_dim64_m24_c1": (64, (2, 4), 1),
        "mid_dim32_m124_c3": (32, (1, 2, 4), 3),
        "tiny_dim8_m12_c3": (8, (1, 2), 3),
    }.items():
        torch.manual_seed(0)
        net = quiet(R.Unet, dim=dim, dim_mults=mults, channels=ch)
        pins["init_sha256"][name] = sha_state(net.state_dict())
        pins["param_count"][name] = sum(p.numel() for p in net.parameters())
        pins.setdefault("state_keys", {})[name] = list(net.state_dict().keys()) if dim <= 32 or name.startswith("cfg2") else None

    # ---- leaf KATs --------------------------------------------------------------------
    g = torch.Generator().manual_seed(1234)
    leaf = {}
    xs = torch.tensor([-25.0, -20.5, -5.0, -1.0, -1e-3, 0.0, 1e-3, 1.0, 5.0, 19.5, 20.0, 20.5, 25.0, 90.0])
    leaf["mish_x"] = np_(xs)
    leaf["mish_y"] = np_(R.Mish()(xs))
    tt = torch.tensor([0, 1, 2, 17, 500, 999])
    for d in (8, 32, 128):
        leaf[f"posemb{d}_t"] = np_(tt)
        leaf[f"posemb{d}_y"] = np_(R.SinusoidalPosEmb(d)(tt))
    xl = torch.randn(2, 16, 4, 4, generator=g)
    ln = R.LayerNorm(16)
    with torch.no_grad():
        ln.g.copy_(torch.randn(1, 16, 1, 1, generator=g)); ln.b.copy_(torch.randn(1, 16, 1, 1, generator=g))
    leaf["ln_x"], leaf["ln_g"], leaf["ln_b"] = np_(xl), np_(ln.g), np_(ln.b)
    leaf["ln_y"] = np_(ln(xl))
    torch.manual_seed(7)
    la = R.LinearAttention(16)
    xa = torch.randn(2, 16, 4, 4, generator=g)
    leaf["la_x"] = np_(xa)
    leaf["la_wqkv"], leaf["la_wout"], leaf["la_bout"] = np_(la.to_qkv.weight), np_(la.to_out.weight), np_(la.to_out.bias)
    leaf["la_y"] = np_(la(xa))
    np.savez_compressed(os.path.join(OUT, "leaf_kats.npz"), **leaf)

    # ---- schedules ----------------------------------------------------------------------
    sched = {}
    for T in (8, 1000):
        gd = R.GaussianDiffusion(torch.nn.Identity(), image_size=(8, 8), timesteps=T)
        for k, v in gd.state_dict().items():
            sched[f"T{T}.{k}"] = np_(v)
    np.savez_compressed(os.path.join(OUT, "schedules.npz"), **sched)

    # ---- tiny UNet: weights + forward + per-leaf captures + grads + sampler -----------------
    torch.manual_seed(0)
    net = quiet(R.Unet, dim=8, dim_mults=(1, 2), channels=3)
    sd = net.state_dict()
    tiny = {"w." + k: np_(v) for k, v in sd.items()}
    x = torch.linspace(-1, 1, 384).reshape(2, 3, 8, 8)
    t = torch.tensor([0, 999])
    caps = {}
    hooks = []
    for name, mod in net.named_modules():
        if name and (len(list(mod.children())) == 0 or isinstance(mod, R.LinearAttention)):
            def mk(nm):
                def hook(m, inp, out):
                    if nm + ".out" not in caps and torch.is_tensor(out):
                        caps[nm + ".in"] = np_(inp[0]); caps[nm + ".out"] = np_(out)
                return hook
            hooks.append(mod.register_forward_hook(mk(name)))
    y = net(x, t)
    for h in hooks:
        h.remove()
    tiny["katA.x"], tiny["katA.t"], tiny["katA.y"] = np_(x), np_(t), np_(y)
    for k, v in caps.items():
        tiny["cap." + k] = v
    # KAT-B: loss + every parameter gradient
    gd = R.GaussianDiffusion(net, image_size=(8, 8), timesteps=1000, loss_type="l1")
    noise = torch.linspace(1, -1, 384).reshape(2, 3, 8, 8)
    net.zero_grad()
    loss = gd.p_losses(x, t, noise)
    loss.backward()
    tiny["katB.noise"] = np_(noise); tiny["katB.loss"] = np_(loss)
    for k, p in net.named_parameters():
        tiny["grad." + k] = np_(p.grad)
    gd2 = R.GaussianDiffusion(net, image_size=(8, 8), timesteps=1000, loss_type="l2")
    net.zero_grad()
    tiny["katB.loss_l2"] = np_(gd2.p_losses(x, t, noise))
    # KAT-C: T=8 sampler; record the noise tape drawn by the reference (randn(shape) x (1+T))
    gd8 = R.GaussianDiffusion(net, image_size=(8, 8), timesteps=8)
    torch.manual_seed(42)
    tape = [torch.randn(2, 3, 8, 8) for _ in range(9)]
    torch.manual_seed(42)
    import tqdm as _tq
    R.tqdm = lambda it, **k: it
    s = gd8.sample(2)
    tiny["katC.tape"] = np.stack([np_(z) for z in tape]); tiny["katC.sample"] = np_(s)
    # 5-step Adam trajectory (lr 1e-4, betas (0.9,0.999)) on fixed (x,t,noise)
    opt = torch.optim.Adam(net.parameters(), lr=1e-4, betas=(0.9, 0.999))
    losses = []
    for _ in range(5):
        opt.zero_grad()
        l = gd.p_losses(x, t, noise)
        l.backward(); opt.step(); losses.append(float(l))
    tiny["adam5.losses"] = np.array(losses, dtype=np.float64)
    tiny["adam5.final.final_conv.1.weight"] = np_(net.final_conv[1].weight)
    tiny["adam5.final.downs.0.0.block1.block.0.weight"] = np_(net.downs[0][0].block1.block[0].weight)
    np.savez_compressed(os.path.join(OUT, "tiny_unet.npz"), **tiny)

    # ---- mid UNet (dim 32, 3 levels, 16x16): seeded weights, forward + loss + grad checks ---
    torch.manual_seed(0)
    net = quiet(R.Unet, dim=32, dim_mults=(1, 2, 4), channels=3)
    gen = torch.Generator().manual_seed(99)
    x = torch.rand(4, 3, 16, 16, generator=gen) * 2 - 1
    t = torch.tensor([0, 10, 500, 999])
    noise = torch.randn(4, 3, 16, 16, generator=gen)
    gd = R.GaussianDiffusion(net, image_size=(16, 16), timesteps=1000)
    mid = {"x": np_(x), "t": np_(t), "noise": np_(noise)}
    mid["y"] = np_(net(x, t))
    net.zero_grad()
    loss = gd.p_losses(x, t, noise)
    loss.backward()
    mid["loss"] = np_(loss)
    for k in ("final_conv.1.weight", "final_conv.0.block.1.weight", "downs.0.0.block1.block.0.weight",
              "downs.1.2.fn.fn.to_qkv.weight", "downs.1.2.fn.norm.g", "mid_attn.fn.fn.to_out.bias",
              "ups.0.3.conv.weight", "downs.0.3.conv.weight", "time_mlp.1.weight", "ups.1.0.res_conv.weight",
              "mid_block1.mlp.1.bias"):
        mid["grad." + k] = np_(dict(net.named_parameters())[k].grad)
    mid["gradnorm_all"] = np.array([float(p.grad.double().norm()) for p in net.parameters()])
    np.savez_compressed(os.path.join(OUT, "mid_unet.npz"), **mid)

    # ---- cfg 2 UNet (dim 128, mults 1-2-4, 32x32), B=2: epsilon prediction + loss ----------
    torch.manual_seed(0)
    net = quiet(R.Unet, dim=128, dim_mults=(1, 2, 4), channels=3)
    gen = torch.Generator().manual_seed(2024)
    x = torch.rand(2, 3, 32, 32, generator=gen) * 2 - 1
    t = torch.tensor([3, 977])
    noise = torch.randn(2, 3, 32, 32, generator=gen)
    gd = R.GaussianDiffusion(net, image_size=(32, 32), timesteps=1000)
    c2 = {"x": np_(x), "t": np_(t), "noise": np_(noise)}
    with torch.no_grad():
        c2["x_noisy"] = np_(gd.q_sample(x, t, noise))
        c2["eps_hat"] = np_(net(gd.q_sample(x, t, noise), t))
    net.zero_grad()
    loss = gd.p_losses(x, t, noise)
    loss.backward()
    c2["loss"] = np_(loss)
    c2["gradnorm_all"] = np.array([float(p.grad.double().norm()) for p in net.parameters()])
    c2["grad.final_conv.1.weight"] = np_(net.final_conv[1].weight.grad)
    c2["grad.time_mlp.3.bias"] = np_(net.time_mlp[3].bias.grad)
    np.savez_compressed(os.path.join(OUT, "cfg2_unet.npz"), **c2)

    with open(os.path.join(OUT, "pins.json"), "w") as f:
        json.dump(pins, f, indent=1)
    for fn in sorted(os.listdir(OUT)):
        print(fn, os.path.getsize(os.path.join(OUT, fn)))


if __name__ == "__main__":
    main()
