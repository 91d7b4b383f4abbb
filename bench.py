#!/usr/bin/env python3
"""Benchmark of the DDPM hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode bf16|fp32] [--batch B]

W warm-up steps (after `--prewarm` set-up steps that absorb the process's cold start), then exactly K timed steps.
A "step" is one full training step of BASELINE cfg 2 (DDPM on CIFAR-10 32x32, UNet base 128,
mults 1-2-4, T=1000, L1 loss, Adam lr 1e-4) on a synthetic batch of B=128 images per GPU that is
already resident in HBM: draw t and eps on the device, q_sample, UNet forward, loss, UNet
backward, gradient all-reduce (N>1, RCCL, overlapped with backward), fused Adam.  Nothing is
skipped or cached between steps.  The denoise rate (hipGraph-replayed p_sample at B=64) is
reported beside it.  One JSON line is printed by rank 0.

For N>1 launch with `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "image-generation-models_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch
import torch.distributed as dist

PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}      # MI355X_MICROARCH.md: dense bf16 MFMA / fp32-input MFMA
TRAIN_GFLOP_PER_IMAGE = 19.424                     # SURVEY.md 8(d), cfg 2: fwd 6.4748 x 3
FWD_GFLOP_PER_IMAGE = 6.4748


def cpu_baseline(seconds_budget: float = 20.0):
    """The CPU oracle (proven equal to the reference in the build container, tests/golden) timed on
    this box's host cores: same UNet/config, fp32, B=16 training steps (fwd+bwd+Adam)."""
    from oracle import ddpm_oracle as O
    threads = min(os.cpu_count() or 1, 64)
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    p = {k: v.requires_grad_(True) for k, v in O.init_unet_params(128, (1, 2, 4), 3).items()}
    tab = O.schedule_tables(1000)
    opt = torch.optim.Adam(list(p.values()), lr=1e-4, betas=(0.9, 0.999))
    B = 16
    x = torch.rand(B, 3, 32, 32) * 2 - 1

    def step():
        t = torch.randint(0, 1000, (B,))
        noise = torch.randn_like(x)
        opt.zero_grad()
        loss, _ = O.p_losses(p, tab, x, t, noise)
        loss.backward()
        opt.step()

    step()
    n, t0 = 0, time.perf_counter()
    while True:
        step(); n += 1
        el = time.perf_counter() - t0
        if el > seconds_budget or n >= 40:
            break
    return {"value": round(B * n / el, 2), "unit": "images/s", "cores": threads, "kind": "port",
            "sample": f"{n} fp32 train steps (fwd+bwd+Adam) of the cfg-2 UNet at B={B} after 1 warm-up step, {el:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--prewarm", type=int, default=25,
                    help="untimed set-up steps before the W warm-up steps: on a fresh box the first ~20 steps of a process are "
                         "CPU-bound (HIP lazy module loads, caching-allocator growth), see tools/coldstart.py")
    ap.add_argument("--mode", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--batch", type=int, default=128, help="images per GPU (reference default 128)")
    ap.add_argument("--denoise-steps", type=int, default=40)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or (os.environ.get("MI_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ)   # the latter: 1-rank RCCL dry run
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from src.models.ddpm import DDPM
    from src.ops import functional as K
    from src.runtime.ddp import FlatGradReducer, broadcast_parameters

    torch.manual_seed(0)
    dm_cfg = {"width": 32, "height": 32, "channels": 3, "transforms": {"normalize": True}}
    model = DDPM(dm_cfg, hidden_dim=128, dim_mults=(1, 2, 4), timesteps=1000, loss_type="l1",
                 lr=1e-4, b1=0.9, b2=0.999).to(dev)                      # configs/model/ddpm.yaml values
    net = model.denoising_model
    net.compute_mode = args.mode
    model.train()
    opt = model.configure_optimizers()
    reducer = None
    if use_dist:
        broadcast_parameters(net.flat_params)
        reducer = FlatGradReducer(net.flat_grads)
        net.grad_ready_hook = reducer.range_ready
        opt.grad_scale = reducer.grad_scale

    B = args.batch
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    imgs = torch.rand(B, 3, 32, 32, device=dev, generator=gen) * 2 - 1    # synthetic [-1,1] batch, HBM resident
    batch = (imgs, None)

    def train_step(i):
        if reducer is not None:
            reducer.begin()
        loss = model.training_step(batch, i)       # randint(t) -> randn(eps) -> q_sample -> UNet -> L1
        loss.backward()                            # UNet backward (+ bucketed RCCL all-reduce)
        if reducer is not None:
            reducer.finish()
        opt.step()                                 # fused Adam over the flat buffer
        return loss

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.prewarm + args.warmup):
        train_step(i)
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = train_step(i)
    sync()
    elapsed = time.perf_counter() - t0
    if use_dist:
        el = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        elapsed = float(el)
    final_loss = float(loss.detach())
    ms_per_step = elapsed / args.steps * 1e3
    images_per_s = world * B * args.steps / elapsed

    # ---- denoise rate: hipGraph-replayed reverse step at B=64 (ddpm.py:520 samples 64 images)
    from src.runtime.sampler import GraphSampler
    model.eval()
    gs = GraphSampler(model.diffusion_model, (64, 3, 32, 32))
    gs._capture()
    gs.x.normal_(); gs.t.fill_(999)
    for _ in range(5):
        gs.z.normal_(); gs.graph.replay()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.denoise_steps):
        gs.z.normal_(); gs.graph.replay()
    sync()
    den = time.perf_counter() - t0
    if use_dist:
        el = torch.tensor([den], device=dev, dtype=torch.float64)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        den = float(el)
    denoise_steps_per_s = world * args.denoise_steps / den
    model.train()

    # ---- roofline of the dominant kernel: HIP events around every conv-family launch of two further
    #      training steps (on the stream the kernels run on); a launch's time is the smaller of its two
    #      measurements, symbols carry the template arguments rocprofv3 prints for the same instantiation
    roof = None
    if rank == 0:
        runs = []
        for i in range(2):
            K.PROBE = []
            train_step(i)
            torch.cuda.synchronize()
            runs.append([(sym, flops, e0.elapsed_time(e1) * 1e-3, desc) for sym, flops, e0, e1, desc in K.PROBE])
        K.PROBE = None
        agg = {}
        if len(runs[0]) == len(runs[1]):
            launches = [(a[0], a[1], min(a[2], b[2])) for a, b in zip(runs[0], runs[1])]
        else:
            launches = [(a[0], a[1], a[2]) for a in runs[1]]
        for sym, flops, sec in launches:
            v = agg.setdefault(sym, [0.0, 0.0, 0])
            v[0] += flops; v[1] += sec; v[2] += 1
        single = {k: v for k, v in agg.items() if "+reduce" not in k}         # symbols that are exactly one kernel
        sym, (fl, sec, cnt) = max(single.items(), key=lambda kv: kv[1][1])
        peak = PEAK_TFLOPS[args.mode]
        ach = fl / sec / 1e12
        traffic, tnote = None, None
        try:                                                                   # PMC passes are separate runs (profiles/)
            with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
                ent = json.load(f).get(sym)
            if ent:
                traffic, tnote = ent["hbm_bytes_per_launch"], ent["note"]
        except OSError:
            pass
        roof = {"kernel": sym, "bound": "mfma", "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s",
                "frac": round(ach / peak, 4), "traffic": traffic, "traffic_note": tnote, "launches_per_step": cnt,
                "avg_launch_us": round(sec / cnt * 1e6, 2), "avg_gflop_per_launch": round(fl / cnt / 1e9, 3),
                "all_conv_kernels": {k: {"tflops": round(v[0] / v[1] / 1e12, 1), "launches_per_step": v[2],
                                         "ms_per_step": round(v[1] * 1e3, 3)} for k, v in sorted(agg.items())}}
    elif world > 1:
        for i in range(2):
            train_step(i)
    sync()

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()

    if rank == 0:
        out = {
            "metric": "ddpm_cifar10_32x32_train_images_per_sec", "value": round(images_per_s, 1), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.mode,
            "data": "synthetic",
            "config": {"workload": "DDPM CIFAR-10 32x32 train step (q_sample+UNet fwd+L1+bwd+allreduce+Adam), "
                                   "UNet base_ch=128 mults 1-2-4, T=1000 (BASELINE configs[1])",
                       "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"dp{world}",
                       "activations": "NHWC; fp32 residual stream, bf16 block- and attention-internal tensors" if args.mode == "bf16" else "fp32 NHWC", "matmul": "bf16 MFMA, fp32 accumulate" if args.mode == "bf16" else "fp32 MFMA"},
            "denoise_steps_per_sec": round(denoise_steps_per_s, 2), "denoise_batch": 64,
            "denoise_image_steps_per_sec": round(denoise_steps_per_s * 64, 1),
            "train_tflops": round(images_per_s * TRAIN_GFLOP_PER_IMAGE / 1e3, 1),
            "denoise_tflops": round(denoise_steps_per_s * 64 * FWD_GFLOP_PER_IMAGE / 1e3, 1),
            "final_loss": round(final_loss, 5),
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
