#!/usr/bin/env python3
"""VQ-VAE (BASELINE cfg 4 shape: CIFAR-10 32x32, latent 64, 512 codes, B=128) training-step throughput on one GPU, with
a per-kernel-family breakdown from HIP events and the oracle's CPU rate beside it.

    python tools/bench_vqvae.py [--batch 128] [--steps 50] [--mode fp32|bf16] [--size 32]
"""
import argparse
import importlib
import json
import os
import sys
import time

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
M = importlib.import_module("image-generation-models_amd.src.models.vqvae")
K = importlib.import_module("image-generation-models_amd.src.ops.functional")


def flops_per_image(size):
    """Forward multiply-adds x 2 of encoder + decoder + codebook distances (training = 3x the conv part)."""
    h2, h4 = size // 2, size // 4
    enc = h2 * h2 * 32 * 3 * 16 + h4 * h4 * 64 * 32 * 16 + h4 * h4 * 64 * 64 * 9 + 3 * h4 * h4 * (64 * 128 * 9 + 128 * 64)
    dec = h4 * h4 * 128 * 64 * 9 + 3 * h4 * h4 * (128 * 128 * 9 + 128 * 128) + h2 * h2 * 64 * 128 * 4 + size * size * 3 * 64 * 4
    vq = h4 * h4 * 64 * 512
    return 2.0 * (enc + dec), 2.0 * vq


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--mode", default="fp32", choices=["fp32", "bf16"])
    ap.add_argument("--size", type=int, default=32)
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--graph", action="store_true", help="capture the step in a hipGraph (src/runtime/graphed.py)")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    dm = {"width": a.size, "height": a.size, "channels": 3, "transforms": {"normalize": True}}
    m = M.VQVAE(dm, encoder={"_target_": "src.networks.vqvae.Encoder"}, decoder={"_target_": "src.networks.vqvae.Decoder"},
                latent_dim=64, lr=1e-3, b1=0.9, b2=0.999, beta=0.25).to(dev)                # configs/model/vqvae.yaml
    m.encoder.compute_mode = m.decoder.compute_mode = a.mode
    m.train()
    opt = m.configure_optimizers()
    imgs = torch.rand(a.batch, 3, a.size, a.size, device=dev) * 2 - 1
    if a.graph:
        OPT = importlib.import_module("image-generation-models_amd.src.runtime.optim")
        G = importlib.import_module("image-generation-models_amd.src.runtime.graphed")
        opt = OPT.FlatAdam(m.flat_nets(), lr=1e-3, betas=(0.9, 0.999), device_state=True)
        gstep = G.GraphedTrainStep(m, opt, (imgs, None))

    def step(i):
        if a.graph:
            return gstep((imgs, None))
        loss = m.training_step((imgs, None), i)
        loss.backward()
        opt.step()
        return loss

    for i in range(a.warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        loss = step(i)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    conv_f, vq_f = flops_per_image(a.size)
    out = {"metric": "vqvae_cifar10_32x32_train_images_per_sec", "value": round(a.batch * a.steps / el, 1), "unit": "images/s",
           "ms_per_step": round(el / a.steps * 1e3, 3), "batch": a.batch, "dtype": a.mode, "final_loss": round(float(loss.detach()), 5),
           "train_tflops": round((3 * conv_f + vq_f) * a.batch * a.steps / el / 1e12, 2)}
    out["graph"] = bool(a.graph)
    if a.graph:
        print(json.dumps(out)); return
    # per-kernel GPU time of one further step (HIP events on the launch stream around every conv-family launch)
    K.PROBE = []
    step(0)
    torch.cuda.synchronize()
    agg = {}
    for sym, fl, e0, e1, *_ in K.PROBE:
        v = agg.setdefault(sym, [0.0, 0.0, 0])
        v[0] += fl; v[1] += e0.elapsed_time(e1) * 1e-3; v[2] += 1
    K.PROBE = None
    out["conv_kernels"] = {k: {"launches": v[2], "ms": round(v[1] * 1e3, 3), "tflops": round(v[0] / v[1] / 1e12, 1)} for k, v in sorted(agg.items())}
    out["conv_ms_per_step"] = round(sum(v[1] for v in agg.values()) * 1e3, 3)
    if a.cpu_steps > 0:                                     # the oracle (torch CPU ops, all host cores) on the same batch
        from oracle import vqvae_oracle as VO
        sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
        x = imgs.cpu()
        VO.training_grads(sd, x, 0.25)
        t0 = time.perf_counter()
        for _ in range(a.cpu_steps):
            VO.training_grads(sd, x, 0.25)
        ce = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(a.batch * a.cpu_steps / ce, 1), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": f"{a.cpu_steps} forward+backward steps of the oracle at B={a.batch} (no optimizer step)"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
