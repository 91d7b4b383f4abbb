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

N>1: one process per GPU.  Under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` the ranks come from
the environment; a bare `python bench.py --gpus N` (no RANK in the environment) re-launches itself under torch.distributed.run
(--standalone, 127.0.0.1) and relays rank 0's JSON line.  `--dry-run-cpu` exercises exactly that launch / rendezvous / max-over-
ranks plumbing with gloo and a stand-in step on the host (no GPU, no kernels; the line says "dry_run": true).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "image-generation-models_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch
import torch.distributed as dist

PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}      # MI355X_MICROARCH.md: dense bf16 MFMA / fp32-input MFMA
PEAK_HBM_GBS = 8000.0                              # HBM3E spec
TRAIN_GFLOP_PER_IMAGE = 19.424                     # SURVEY.md 8(d), cfg 2: fwd 6.4748 x 3
FWD_GFLOP_PER_IMAGE = 6.4748
NAMED_GFLOP = 38.65                                # level-0 fused GN-apply+Mish(+temb)+Conv3x3 [128,32,32,128]->128
NAMED_MB = {"fp32": 134.9, "bf16": 67.5}           # its algorithmic HBM bytes by activation storage


# kernel symbol -> the source file that implements it (roofline.traffic comes from PMC passes made in another run: the JSON records the
# hash of this file at collection time, bench.py compares it with the tree's and says `traffic_stale` when the kernel changed since)
KERNEL_SOURCES = {"conv_pw_kernel": "conv_pw.hip", "conv1x1_pw_kernel": "conv_pw.hip", "conv3x3_halo_kernel": "conv3x3_halo.hip",
                  "wgrad_tr_kernel": "wgrad_tr.hip", "wgrad1x1_tr_kernel": "wgrad1x1_tr.hip", "gn_mish": "norm_act.hip",
                  "chan_ln": "norm_act.hip", "linattn": "linattn.hip", "igemm": "igemm_conv.hip", "conv_gt_kernel": "conv_pw.hip", "wgrad_s2": "wgrad_s2_tr.hip",
                  "adam_kernel": "elementwise.hip", "pack_weights": "conv3x3_halo.hip", "small_c": "small_channel.hip", "cvt_colsum": "elementwise.hip",
                  "colsum": "elementwise.hip", "small_gemm": "small_gemm.hip", "partial_sum": "small_channel.hip", "eps_loss": "elementwise.hip",
                  "rowsum_batch": "elementwise.hip", "q_sample": "elementwise.hip"}


def kernel_source_sha(sym: str):
    """(relative path, sha256) of the csrc file behind a kernel symbol, or (None, None)"""
    import hashlib
    for key, fn in KERNEL_SOURCES.items():
        if sym.startswith(key):
            rel = os.path.join("image-generation-models_amd", "csrc", fn)
            try:
                with open(os.path.join(ROOT, rel), "rb") as f:
                    return rel, hashlib.sha256(f.read()).hexdigest()
            except OSError:
                return rel, None
    return None, None


def cpu_baseline(budget_s: float = 28.0):
    """The CPU oracle (proven equal to the reference in the build container, tests/golden) timed on this box's host cores,
    fp32, the three legs of BASELINE.md section 4: train steps (fwd+bwd+Adam) at B=16 and at B=128 (the GPU line's batch), one
    p_sample at B=64.  Each leg: 1 warm-up call, then timed calls until its share of the budget is spent (at least 1)."""
    from oracle import ddpm_oracle as O
    threads = min(os.cpu_count() or 1, 64)
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    p = {k: v.requires_grad_(True) for k, v in O.init_unet_params(128, (1, 2, 4), 3).items()}
    tab = O.schedule_tables(1000)
    opt = torch.optim.Adam(list(p.values()), lr=1e-4, betas=(0.9, 0.999))

    def timed(fn, share, max_calls):
        fn()
        n, t0 = 0, time.perf_counter()
        while True:
            fn(); n += 1
            el = time.perf_counter() - t0
            if el > share or n >= max_calls:
                return n, el

    def train_leg(B, share, max_calls):
        x = torch.rand(B, 3, 32, 32) * 2 - 1

        def step():
            t = torch.randint(0, 1000, (B,))
            noise = torch.randn_like(x)
            opt.zero_grad()
            loss, _ = O.p_losses(p, tab, x, t, noise)
            loss.backward()
            opt.step()
        n, el = timed(step, share, max_calls)
        return {"value": round(B * n / el, 2), "unit": "images/s", "batch": B, "calls": n, "seconds": round(el, 1)}

    def sample_leg(B, share, max_calls):
        x = torch.randn(B, 3, 32, 32)
        t = torch.full((B,), 500, dtype=torch.long)
        pd = {k: v.detach() for k, v in p.items()}

        def step():
            with torch.no_grad():
                O.p_sample_update(tab, x, t, O.unet_forward(pd, x, t), torch.randn_like(x))
        n, el = timed(step, share, max_calls)
        return {"value": round(n / el, 3), "unit": "denoise steps/s", "batch": B, "calls": n, "seconds": round(el, 1),
                "image_steps_per_sec": round(B * n / el, 1)}

    legs = {"train_b16": train_leg(16, budget_s * 0.2, 8), "train_b128": train_leg(128, budget_s * 0.3, 3),
            "p_sample_b64": sample_leg(64, budget_s * 0.1, 3)}
    main = legs["train_b128"]
    return {"value": main["value"], "unit": "images/s", "cores": threads, "kind": "port",
            "sample": f"{main['calls']} fp32 train steps (fwd+bwd+Adam) of the cfg-2 UNet at B=128 (the GPU line's batch) after 1 warm-up "
                      f"step, {main['seconds']} s; legs: B=16 train, B=128 train, one p_sample at B=64 (BASELINE.md section 4)",
            "legs": legs}


def relaunch(args, extra_env=None, capture=False, drop=()):
    """`python bench.py --gpus N` without a launcher: run N ranks of this script under torch.distributed.run and relay the line."""
    port = 29500 + ((os.getpid() + (hash(str(extra_env)) % 97)) % 400)
    argv = [a_ for a_ in sys.argv[1:] if a_ not in drop]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MI_BENCH_SELF_LAUNCHED"] = "1"
    env.update(extra_env or {})
    if not capture:
        return subprocess.call(cmd, env=env)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r.returncode, (json.loads(lines[-1]) if lines else None), r.stderr[-2000:]


def sweep_channels(args):
    """--sweep-channels: the whole N-rank bench once per NCCL_MAX_NCHANNELS in {default, 8, 16} (RCCL reads the variable when the
    communicator is created, so every value is its own set of processes); ONE line: the default run's, with the three results under
    "channel_sweep".  Only from the self-launching form (`python bench.py --gpus N --sweep-channels`, no launcher around it)."""
    res, main_line = [], None
    for val in (None, "8", "16"):
        env = {} if val is None else {"NCCL_MAX_NCHANNELS": val}
        if val is None:
            env["MI_BENCH_UNSET_NCHANNELS"] = "1"
        rc, line, err = relaunch(args, env, capture=True, drop=("--sweep-channels",))
        if rc != 0 or line is None:
            res.append({"NCCL_MAX_NCHANNELS": val or "default", "error": err[-400:]})
            continue
        res.append({"NCCL_MAX_NCHANNELS": val or "default", "value": line.get("value"), "ms_per_step": line.get("ms_per_step"),
                    "ways_ms": (line.get("comm") or {}).get("ways_ms")})
        if main_line is None:
            main_line = line
    if main_line is None:
        print(json.dumps({"error": "every run of the channel sweep failed", "channel_sweep": res}))
        return 1
    main_line["channel_sweep"] = res
    print(json.dumps(main_line))
    return 0


def rank_spread(elapsed_s, steps, world, device="cpu"):
    """ms per step of every rank (all_gather): min / max tell a straggler from a uniformly slow step."""
    if world <= 1:
        return {"min": round(elapsed_s / steps * 1e3, 3), "max": round(elapsed_s / steps * 1e3, 3)}
    t = torch.tensor([elapsed_s / steps * 1e3], device=device, dtype=torch.float64)
    out = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    v = [float(o) for o in out]
    return {"min": round(min(v), 3), "max": round(max(v), 3), "per_rank": [round(x, 3) for x in v]}


def timed_leg(step, steps, warmup, sync, world, device="cpu"):
    """warmup + `steps` calls of step(i), bracketed by sync() (barrier + device synchronize); MAX over ranks; ms per step."""
    for i in range(warmup):
        step(i)
    sync()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    sync()
    el = torch.tensor([time.perf_counter() - t0], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    return float(el) / steps * 1e3


def dry_run_cpu(args, world, rank):
    """Launch plumbing on the host: gloo ranks, a stand-in step (bucketed all-reduce over a small flat buffer + a sleep),
    barrier-bracketed timing, MAX over ranks, rank 0 prints the line.  No kernels run; nothing here is a measurement -- but the
    rank check, the per-rank spread, the three-ways comparison, the one-rank data-parallel leg and the channel sweep's relaunching
    are the code the GPU run executes."""
    from src.runtime.ddp import FlatGradReducer
    solo_group = False
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    elif not args.no_extras:                               # N = 1 through the data-parallel code path (a one-rank group)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 400))
        dist.init_process_group("gloo", rank=0, world_size=1)
        solo_group = True
    grouped = world > 1 or solo_group
    flat = torch.ones(1 << 16)
    red = FlatGradReducer(flat, bucket_bytes=1 << 16) if grouped else None
    ones = torch.ones(1)
    if grouped:
        dist.all_reduce(ones)
    if int(ones.item()) != world:
        raise SystemExit(f"bench.py: the backend connected {int(ones.item())} ranks, --gpus says {world}")

    def step(i=0):
        if red is not None:
            red.begin()
            for hi in range(flat.numel(), 0, -(1 << 13)):
                red.range_ready(hi - (1 << 13), hi)
            red.finish()
        time.sleep(0.002)

    def sync():
        if world > 1:
            dist.barrier()
    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    mine = time.perf_counter() - t0
    spread = rank_spread(mine, args.steps, world)
    el = torch.tensor([mine], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    ways = None
    if red is not None:
        ways = {"segmented_graph": round(float(el) / args.steps * 1e3, 3), "eager_overlap": round(timed_leg(step, 2, 1, sync, world), 3)}
        red.defer = True
        ways["eager_allreduce_after_backward"] = round(timed_leg(step, 2, 1, sync, world), 3)
        red.defer = False
        ways["exposed_comm_ms"] = round(ways["eager_allreduce_after_backward"] - ways["eager_overlap"], 3)
    if rank == 0:
        print(json.dumps({"metric": "ddpm_cifar10_32x32_train_images_per_sec", "dry_run": True, "value": None, "unit": "images/s",
                          "n_gpus": world, "rccl_ranks": int(ones.item()), "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(float(el) / args.steps * 1e3, 3), "scaling": "weak", "higher_is_better": True,
                          "rank_ms_per_step": spread,
                          "buckets_bytes": [4 * (hi - lo) for lo, hi in red.launched] if red else [],
                          "comm": {"ways_ms": ways, "NCCL_MAX_NCHANNELS": os.environ.get("NCCL_MAX_NCHANNELS")} if red else None,
                          "dp_path_n1": ({"ms_per_step": ways["segmented_graph"], "rccl_ranks": int(ones.item())} if solo_group else None),
                          "self_launched": os.environ.get("MI_BENCH_SELF_LAUNCHED") == "1"}))
    if grouped:
        dist.destroy_process_group()


def named_kernel_line(K, dev, N=128, H=32, W=32, Cc=128):
    """north_star's named unit at its canonical shape: GroupNorm-apply + Mish (+ time bias) + Conv3x3 on
    X = [128,32,32,128] NHWC -> 128 channels, level 0 of cfg 2, for both activation storages of the bf16-MFMA mode.  Every entry's
    `unit_us` is EVERYTHING the unit needs, timed with HIP events on the launch stream (10 calls back to back per sample, median of 7 samples): the two-pass unit = GroupNorm
    kernel + conv; a fused unit = the fused conv + whatever produces its statistics (a statistics pass over the tensor, or what the
    sums cost the PRODUCING conv's epilogue: that conv timed with and without them); the single fused launch is reported beside it as
    `fused_launch_us`.  hbm_frac = algorithmic bytes / unit time / 8 TB/s, mfma_frac = 38.65 GFLOP / unit time / 2.5 PFLOP/s."""
    canonical = (N, H, W, Cc) == (128, 32, 32, 128)
    gflop = NAMED_GFLOP if canonical else round(2.0 * N * H * W * Cc * Cc * 9 / 1e9, 2)
    # algorithmic bytes: x in + y out in the storage type, the bf16 weights, the [3][N][C] coefficients
    mb = NAMED_MB if canonical else {k: round((N * H * W * Cc * 2 * e + 9 * Cc * Cc * 2 + 3 * N * Cc * 4) / 1e6, 1) for k, e in (("fp32", 4), ("bf16", 2))}
    out = {"shape": f"[{N},{H},{W},{Cc}] NHWC -> {Cc}, 3x3/s1/p1", "gflop": gflop}
    sums_ok = (Cc // 8) % 16 == 0            # the epilogue's sums are per 16-channel slab: groups of >= 16 channels
    g = torch.Generator(device=dev).manual_seed(7)
    gamma = torch.ones(Cc, device=dev); beta = torch.zeros(Cc, device=dev)
    temb = torch.randn(N, Cc, device=dev, generator=g) * 0.1
    w32 = torch.randn(9 * Cc * Cc, device=dev, generator=g) * 0.03                 # master layout [tap][ci][co]
    table, nent, tiles = K.pack_table([(0, 9, Cc, Cc)], dev)
    wd, w, wdq, wq = (torch.zeros(w32.numel(), device=dev, dtype=torch.bfloat16) for _ in range(4))
    K.pack_weights_bf16(table, nent, tiles, w32, wd, w, wdq, wq)
    bias = torch.zeros(Cc, device=dev)

    def timed(fns, reps=10, rounds=7):
        """{name: callable} -> us per call of each callable, and the kernel symbols it launches.  A call is timed the way it runs inside the
        replayed step -- back to back on a busy stream: `reps` calls between two HIP events on the launch stream, `rounds` interleaved
        rounds over the legs, the median round.  (Until round 5 every call was synchronised on its own: an idle chip in front of each
        launch reads 3 - 5 us more per launch than the same kernel shows in the step's rocprofv3 trace.)"""
        syms = {}
        for k, fn in fns.items():
            K.PROBE = []
            fn()
            torch.cuda.synchronize()
            syms[k] = [q[0] for q in K.PROBE]
            K.PROBE = None
            for _ in range(2):
                fn()
        t = {k: [] for k in fns}
        for _ in range(rounds):
            for k, fn in fns.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _r in range(reps):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                t[k].append(e0.elapsed_time(e1) * 1e3 / reps)
        return {k: sorted(v)[len(v) // 2] for k, v in t.items()}, syms

    for sto, dt in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
        x = torch.randn(N, H, W, Cc, device=dev, generator=g).to(dt)           # c1: the raw output of block1's conv
        xin = torch.randn(N, H, W, Cc, device=dev, generator=g).to(dt)         # block1's input (producer timing)
        xs = x.float().view(N, H * W, Cc // 16, 16)
        sums = K.gn_sums_encode(torch.stack([xs.sum((1, 3)), (xs * xs).sum((1, 3))], dim=-1))       # what c1's producer would have left
        scratch = torch.zeros_like(sums)                                       # the timed producer adds into this one
        gn = (sums, gamma, beta, temb, 8, 1e-5)
        if K.conv3x3_gn_mish(x, K.gn_stats_coef(x, gamma, beta, temb=temb)[1], w, K=Cc, Nc=Cc, bias=bias, wq=wq) is None:
            out[sto + "_storage"] = {"unsupported": "no fused GroupNorm + Mish + conv kernel takes this shape"}
            continue
        stats, coef = K.gn_stats_coef(x, gamma, beta, temb=temb)
        hbuf = [None]

        def gn_pass():
            hbuf[0], _ = K.gn_mish_fwd(x, gamma, beta, temb=temb, out_dtype=dt)
        gn_pass()
        legs = {
            "gn": gn_pass,
            "conv": lambda: K.conv3x3_bf16w(hbuf[0], w, K=Cc, Nc=Cc, flip=False, ksize=3, bias=bias, out_dtype=dt, wq=wq),
            "stats": lambda: K.gn_stats_coef(x, gamma, beta, temb=temb),
            "fused_coef": lambda: K.conv3x3_gn_mish(x, coef, w, K=Cc, Nc=Cc, bias=bias, wq=wq),
        }
        if sums_ok:
            legs.update({
                "fused_sums": lambda: K.conv3x3_gn_mish(x, None, w, K=Cc, Nc=Cc, bias=bias, gn=gn, wq=wq),
                "producer": lambda: K.conv3x3_bf16w(xin, w, K=Cc, Nc=Cc, flip=False, ksize=3, bias=bias, out_dtype=dt, wq=wq),
                "producer_sums": lambda: K.conv3x3_bf16w(xin, w, K=Cc, Nc=Cc, flip=False, ksize=3, bias=bias, out_dtype=dt, gn_sums=scratch, wq=wq)})
        md, syms = timed(legs)

        def line(us):
            return {"unit_us": round(us, 2), "algorithmic_mb": mb[sto],
                    "hbm_gbs": round(mb[sto] * 1e6 / (us * 1e-6) / 1e9, 1),
                    "hbm_frac": round(mb[sto] * 1e6 / (us * 1e-6) / 1e9 / PEAK_HBM_GBS, 4),
                    "tflops": round(gflop * 1e9 / (us * 1e-6) / 1e12, 1),
                    "mfma_frac": round(gflop * 1e9 / (us * 1e-6) / 1e12 / PEAK_TFLOPS["bf16"], 4)}
        sums_cost = max(md["producer_sums"] - md["producer"], 0.0) if sums_ok else 0.0
        ent = {"two_pass": dict(launches_us={syms["gn"][0]: round(md["gn"], 2), syms["conv"][0]: round(md["conv"], 2)},
                                **line(md["gn"] + md["conv"])),
               # statistics by a pass over c1 (mi_gn_stats_coef), then the fused conv fed by the coefficient tensor
               "fused": dict(kernel=syms["fused_coef"][-1], fused_launch_us=round(md["fused_coef"], 2),
                             statistics_pass_us=round(md["stats"], 2), **line(md["fused_coef"] + md["stats"])),
               # statistics from the producing conv's epilogue, coefficients resolved inside the fused kernel: Block -> Block is two
               # launches; the unit = the fused launch + what the sums add to the producer
               "fused_epilogue_stats": (dict(kernel=syms["fused_sums"][-1], fused_launch_us=round(md["fused_sums"], 2),
                                             producer_conv=syms["producer_sums"][-1], producer_conv_us=round(md["producer"], 2),
                                             producer_conv_with_sums_us=round(md["producer_sums"], 2),
                                             statistics_cost_us=round(sums_cost, 2), **line(md["fused_sums"] + sums_cost))
                                        if sums_ok else {"unsupported": "groups of fewer than 16 channels: no per-slab sums"})}
        out[sto + "_storage"] = ent
    if not canonical:
        return out
    out["default_path"] = ("bf16 storage; inference: fused_epilogue_stats wherever the private-weight-stream kernel takes block2's conv "
                           "(Unet.fuse_gn_conv = 'auto'), training: two_pass (block2's weight gradient reads the normalised tensor)")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--prewarm", type=int, default=25,
                    help="untimed set-up steps before the W warm-up steps: on a fresh box the first ~20 steps of a process are "
                         "CPU-bound (HIP lazy module loads, caching-allocator growth), see tools/coldstart.py")
    ap.add_argument("--mode", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--batch", type=int, default=128, help="images per GPU (reference default 128)")
    ap.add_argument("--denoise-steps", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the fp32-mode leg, the named-kernel microbenchmark and the CPU baseline")
    ap.add_argument("--graph", type=int, default=int(os.environ.get("MI_BENCH_GRAPH", "-1")),
                    help="1: replay the training step as a hipGraph (under data parallelism: a chain of graphs cut at the gradient "
                         "buckets), 0: eager, -1 (default): the Trainer's default = replay, eager if the capture fails")
    ap.add_argument("--dry-run-cpu", action="store_true", help="launch/rendezvous plumbing only (gloo, host stand-in step)")
    ap.add_argument("--cfg", type=int, default=2, choices=[2, 3],
                    help="2 (default, the metric's configuration): CIFAR-10 32x32, UNet 128 / 1-2-4; 3: BASELINE configs[2], CelebA 64x64, "
                         "UNet 64 / 1-2-4-8 (use --batch 32 for its per-GPU batch at 8 GPUs); the line's metric name and workload say which")
    ap.add_argument("--sweep-channels", action="store_true",
                    help="self-launching form only: run the N-rank bench for NCCL_MAX_NCHANNELS in {default, 8, 16} and report all three")
    args = ap.parse_args()

    if os.environ.get("MI_BENCH_UNSET_NCHANNELS") == "1":
        os.environ.pop("NCCL_MAX_NCHANNELS", None)
    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(sweep_channels(args) if args.sweep_channels else relaunch(args))
    if args.sweep_channels and not args.dry_run_cpu:
        print("[bench] --sweep-channels needs the self-launching form (python bench.py --gpus N --sweep-channels, N > 1); ignored", file=sys.stderr)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.dry_run_cpu:
        return dry_run_cpu(args, world, rank)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or (os.environ.get("MI_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ)   # the latter: 1-rank RCCL dry run
    rccl_ranks = 1
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)                           # how many ranks RCCL really connected
        rccl_ranks = int(ones.item())
        if rccl_ranks != world:                         # a scaling number over fewer ranks than it claims is worse than no number
            raise SystemExit(f"bench.py: RCCL connected {rccl_ranks} ranks, --gpus says {world}")

    from src.models.ddpm import DDPM
    from src.ops import functional as K
    from src.runtime.ddp import FlatGradReducer, broadcast_parameters

    SIDE = 32 if args.cfg == 2 else 64
    TRAIN_GF = TRAIN_GFLOP_PER_IMAGE if args.cfg == 2 else 26.278
    FWD_GF = FWD_GFLOP_PER_IMAGE if args.cfg == 2 else 8.7593

    def build(mode, cfg=None):
        cfg = cfg or args.cfg
        side = 32 if cfg == 2 else 64
        torch.manual_seed(0)
        dm_cfg = {"width": side, "height": side, "channels": 3, "transforms": {"normalize": True}}
        m = DDPM(dm_cfg, hidden_dim=128 if cfg == 2 else 64, dim_mults=(1, 2, 4) if cfg == 2 else (1, 2, 4, 8), timesteps=1000,
                 loss_type="l1", lr=1e-4, b1=0.9, b2=0.999).to(dev)      # configs/model/ddpm.yaml values
        m.denoising_model.compute_mode = mode
        m.train()
        return m, m.denoising_model, m.configure_optimizers()

    model, net, opt = build(args.mode)
    reducer = None
    bucket_mb = int(os.environ.get("MI_DDP_BUCKET_MB", "32"))
    if use_dist:
        broadcast_parameters(net.flat_params)
        reducer = FlatGradReducer(net.flat_grads, bucket_bytes=bucket_mb << 20)
        net.grad_ready_hook = reducer.range_ready
        opt.grad_scale = reducer.grad_scale

    B = args.batch
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    imgs = torch.rand(B, 3, SIDE, SIDE, device=dev, generator=gen) * 2 - 1    # synthetic [-1,1] batch, HBM resident
    batch = (imgs, None)

    def eager_step(i, m=None, o=None):
        m, o = m or model, o or opt
        if reducer is not None:
            reducer.begin()
        loss = m.training_step(batch, i)           # randint(t) -> randn(eps) -> q_sample -> UNet -> L1
        loss.backward()                            # UNet backward (+ bucketed RCCL all-reduce)
        if reducer is not None:
            reducer.finish()
        o.step()                                   # fused Adam over the flat buffer
        return loss

    # hipGraph replay of the step: same kernels, same order.  One graph launch per step in a single process; under data parallelism
    # a chain of graphs cut at the gradient buckets with the all-reduces issued between them (src/runtime/graphed.py)
    use_graph = args.graph != 0
    graph_note = None
    train_step = eager_step
    if use_graph:
        from src.runtime.graphed import GraphedTrainStep, SegmentedGraphedTrainStep
        opt.device_state = True
        for i in range(3):
            eager_step(i)
        try:
            gstep = SegmentedGraphedTrainStep(model, opt, reducer, batch) if reducer is not None else GraphedTrainStep(model, opt, batch, warmup=0)
            train_step = lambda i: gstep(batch)        # noqa: E731
        except Exception as exc:                       # noqa: BLE001  (what Trainer.fit does in its automatic mode)
            if args.graph == 1:
                raise
            use_graph, graph_note = False, f"capture failed ({type(exc).__name__}: {exc}); eager"
            torch.cuda.synchronize()
    if use_dist:                                       # every rank must run the same kind of step
        flag = torch.tensor([1 if use_graph else 0], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if use_graph and int(flag) == 0:
            use_graph, graph_note, train_step = False, "capture failed on another rank; eager", eager_step

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
    spread = rank_spread(elapsed, args.steps, world if use_dist else 1, dev)
    if use_dist:
        el = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        elapsed = float(el)
    # the shader clock right behind the timed steps (the chip is in its sustained state: a probe after an idle stretch reads 10 - 15 % low)
    sclk = None
    if rank == 0 and not args.no_extras:
        try:
            sclk = K.clock_probe(dev, usec=300)
        except Exception:                                   # noqa: BLE001
            sclk = None
    final_loss = float(loss.detach())
    # a throughput of steps that trained garbage is not a measurement: the loss, every weight and every gradient after the timed steps
    # must be finite (outside the timed region; round 5 found a replayed-graph fault that left NaN weights and a normal step time)
    unet_ = model.denoising_model
    state_finite = bool(torch.isfinite(unet_.flat_params).all()) and bool(torch.isfinite(unet_.flat_grads).all()) and final_loss == final_loss \
        and abs(final_loss) != float("inf")
    if use_dist:                                        # every rank leaves together (a lone SystemExit would hang the peers' collectives)
        flag_ = torch.tensor([1 if state_finite else 0], device=dev)
        dist.all_reduce(flag_, op=dist.ReduceOp.MIN)
        state_finite = bool(int(flag_))
    if not state_finite:
        raise SystemExit(f"bench.py: non-finite training state after the timed steps (loss {final_loss}); no throughput is reported")
    ms_per_step = elapsed / args.steps * 1e3
    images_per_s = world * B * args.steps / elapsed

    # ---- exposed all-reduce time: GPU time on the compute stream between the end of backward and the start of Adam
    comm = None
    if reducer is not None:
        waits = []
        for i in range(5):
            reducer.begin()
            loss = model.training_step(batch, i); loss.backward()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); reducer.finish(); e1.record()
            opt.step()
            waits.append((e0, e1))
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in waits)
        comm = {"allreduce_ms_exposed": round(ms[len(ms) // 2], 3), "buckets_bytes": [4 * (hi - lo) for lo, hi in reducer.launched],
                # the knobs that decide how much of the chip the collectives take from backward
                "bucket_mb": bucket_mb, "NCCL_MAX_NCHANNELS": os.environ.get("NCCL_MAX_NCHANNELS"),
                "NCCL_MIN_NCHANNELS": os.environ.get("NCCL_MIN_NCHANNELS"),
                "graph_segments": len(gstep.segments) + 1 if (use_graph and reducer is not None) else None}
        if use_dist:
            t = torch.tensor([comm["allreduce_ms_exposed"]], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            comm["allreduce_ms_exposed"] = round(float(t), 3)
        # the step three ways, so that the exposed communication is a subtraction and not an inference: the timed leg above
        # (segmented graph when it captured), eager with the all-reduces overlapped with backward, eager with the SAME buckets issued
        # after backward
        w_ = world if use_dist else 1
        ways = {"segmented_graph": round(ms_per_step, 3) if use_graph else None,
                "eager_overlap": round(timed_leg(eager_step, 20, 3, sync, w_, dev), 3)}
        reducer.defer = True
        ways["eager_allreduce_after_backward"] = round(timed_leg(eager_step, 20, 3, sync, w_, dev), 3)
        reducer.defer = False
        ways["exposed_comm_ms"] = round(ways["eager_allreduce_after_backward"] - ways["eager_overlap"], 3)
        comm["ways_ms"] = ways

    # ---- denoise rate: hipGraph-replayed reverse step at B=64 (ddpm.py:520 samples 64 images)
    from src.runtime.sampler import GraphSampler
    model.eval()
    gs = GraphSampler(model.diffusion_model, (64, 3, SIDE, SIDE))
    gs._capture()
    gs.set_image(torch.randn_like(gs.x)); gs.t.fill_(999)
    # (60 untimed replays first: from idle the chip's shader clock takes tens of milliseconds of load to reach its sustained value --
    #  tools/proto/clock_probe.py: 2 050 MHz in the first millisecond, 2 390 after ~30 ms -- and a T = 1000 sampling run lasts a second)
    for _ in range(60):
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

    # ---- per-kernel times of the step: HIP events (on the launch stream) around every library launch of three further EAGER
    #      training steps; a launch's time is the smallest of its measurements; symbols carry the template arguments rocprofv3
    #      prints.  The contraction and the partial-tile reduce of the weight-gradient kernel are timed as separate launches.
    roof = None
    if rank == 0:
        runs = []
        for i in range(3):
            K.PROBE = []
            eager_step(i)
            torch.cuda.synchronize()
            runs.append([(sym, fl, e0.elapsed_time(e1) * 1e-3, desc, nb) for sym, fl, e0, e1, desc, nb in K.PROBE])
        K.PROBE = None
        if len({len(r) for r in runs}) == 1:
            launches = [(a[0], a[1], min(x[2] for x in grp), a[4]) for grp in zip(*runs) for a in (grp[0],)]
        else:
            launches = [(a[0], a[1], a[2], a[4]) for a in runs[-1]]
        agg = {}
        for sym, fl, sec, nb in launches:
            v = agg.setdefault(sym, [0.0, 0.0, 0, 0.0])
            v[0] += fl; v[1] += sec; v[2] += 1; v[3] += nb
        sym, (fl, sec, cnt, nb) = max(agg.items(), key=lambda kv: kv[1][1])          # the symbol with the most time per step
        peak = PEAK_TFLOPS[args.mode]
        traffic, tnote, tstale = None, None, None
        for name in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):   # PMC passes are separate runs (profiles/)
            try:
                with open(os.path.join(ROOT, "profiles", name)) as f:
                    ent = json.load(f).get(sym)
                if ent:
                    traffic, tnote = ent["hbm_bytes_per_launch"], ent["note"] + f" ({name})"
                    # stale = the kernel's source is not the one the counters were collected on (no recorded hash: unknown -> stale)
                    tstale = ent.get("source_sha256") is None or ent.get("source_sha256") != kernel_source_sha(sym)[1]
                    break
            except OSError:
                pass
        mfma_bound = fl > 0 and fl / max(nb, 1.0) > 312.0                            # bf16 machine balance, SURVEY 8(d)
        ach = fl / sec / 1e12 if mfma_bound else nb / sec / 1e9
        pk = peak if mfma_bound else PEAK_HBM_GBS
        table = {}
        for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            ent = {"launches_per_step": v[2], "ms_per_step": round(v[1] * 1e3, 3)}
            if v[0] > 0:
                ent["tflops"] = round(v[0] / v[1] / 1e12, 1)
            if v[3] > 0:
                ent["algorithmic_gbs"] = round(v[3] / v[1] / 1e9, 1)
            table[k] = ent
        roof = {"kernel": sym, "bound": "mfma" if mfma_bound else "hbm", "achieved": round(ach, 1), "peak": pk,
                "unit": "TFLOP/s" if mfma_bound else "GB/s", "frac": round(ach / pk, 4), "traffic": traffic, "traffic_stale": tstale, "traffic_note": tnote,
                "launches_per_step": cnt, "avg_launch_us": round(sec / cnt * 1e6, 2),
                "avg_gflop_per_launch": round(fl / cnt / 1e9, 3), "avg_algorithmic_mb_per_launch": round(nb / cnt / 1e6, 2),
                "probed_ms_per_step": round(sum(v[1] for v in agg.values()) * 1e3, 3),
                "all_kernels": table}
        if not args.no_extras and args.cfg == 2:
            roof["named_kernel"] = named_kernel_line(K, dev)
            # ... and at the one shape of the benchmarked configurations that SURVEY 8(d)(iv) calls HBM-bound: cfg 3's first level
            try:
                roof["named_kernel_cfg3_level0"] = named_kernel_line(K, dev, 32, 64, 64, 64)
            except Exception as exc:                        # noqa: BLE001  (an extra entry: reported, never fatal)
                roof["named_kernel_cfg3_level0"] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
    elif world > 1:
        for i in range(3):
            eager_step(i)
    sync()

    # ---- N = 1 through the data-parallel code path (one-rank RCCL group, bucketed reducer, segmented graph): what SCALE's N = 1
    #      line would read if the driver launched it under torch.distributed.run, next to the plain single-process number above
    dp_path = None
    if world == 1 and not use_dist and not args.no_extras:
        # (RCCL prints a version banner through C stdio when its first communicator comes up: this leg's stdout goes to stderr, so that
        #  the N = 1 run keeps printing exactly one line)
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            from src.runtime.graphed import SegmentedGraphedTrainStep
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 400))
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
            one = torch.ones(1, device=dev); dist.all_reduce(one)
            red1 = FlatGradReducer(net.flat_grads, bucket_bytes=bucket_mb << 20)
            net.grad_ready_hook = red1.range_ready
            opt.device_state = True
            g1 = SegmentedGraphedTrainStep(model, opt, red1, batch)
            ms1 = timed_leg(lambda i: g1(batch), 30, 12, torch.cuda.synchronize, 1, dev)
            dp_path = {"value": round(B / ms1 * 1e3, 1), "ms_per_step": round(ms1, 3), "rccl_ranks": int(one.item()),
                       "graph_segments": len(g1.segments) + 1, "buckets_bytes": [4 * (hi - lo) for lo, hi in red1.launched]}
            del g1
        except Exception as exc:                            # noqa: BLE001  (an extra: reported, never fatal)
            dp_path = {"error": f"{type(exc).__name__}: {exc}"[:300]}
        finally:
            net.grad_ready_hook = None
            if dist.is_initialized():
                dist.destroy_process_group()
            try:
                import ctypes
                ctypes.CDLL(None).fflush(None)
            except Exception:       # noqa: BLE001
                pass
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    # ---- the parity-carrying fp32 mode (exact-fp32 MFMA), a short driver-timed leg
    fp32_mode = None
    if rank == 0 and world == 1 and not args.no_extras and args.mode != "fp32":
        del gs
        m32, n32, o32 = build("fp32")
        for i in range(6):
            eager_step(i, m32, o32)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 10
        for i in range(n):
            eager_step(i, m32, o32)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        # the three symbols with the most time in that step (HIP events on the launch stream around every launch of one more step)
        K.PROBE = []
        try:
            eager_step(0, m32, o32)
            torch.cuda.synchronize()
            agg32 = {}
            for sym, fl, e0, e1, _, nb in K.PROBE:
                v = agg32.setdefault(sym, [0.0, 0.0, 0])
                v[0] += fl; v[1] += e0.elapsed_time(e1) * 1e-3; v[2] += 1
        finally:
            K.PROBE = None
        top32 = {k: {"launches_per_step": v[2], "ms_per_step": round(v[1] * 1e3, 3), "tflops": round(v[0] / v[1] / 1e12, 1),
                     "frac_of_fp32_mfma_peak": round(v[0] / v[1] / 1e12 / PEAK_TFLOPS["fp32"], 3)}
                 for k, v in sorted(agg32.items(), key=lambda kv: -kv[1][1])[:3] if v[1] > 0}
        fp32_mode = {"value": round(B * n / el, 1), "unit": "images/s", "ms_per_step": round(el / n * 1e3, 3), "steps": n,
                     "train_tflops": round(B * n / el * TRAIN_GF / 1e3, 1), "peak_tflops": PEAK_TFLOPS["fp32"],
                     "top_kernels": top32,
                     "note": "exact-fp32 MFMA mode (v_mfma_f32_32x32x2_f32): the mode that carries the <=1e-4 epsilon-prediction bar"}
        del m32, n32, o32

    # ---- BASELINE configs[2] at its per-GPU batch (CelebA 64x64, UNet 64 / 1-2-4-8, B = 256 / 8 GPUs = 32): a short graph-replayed leg,
    #      so that the driver's record carries a cfg-3 number too (its own line: python bench.py --cfg 3 --batch 32)
    cfg3 = None
    if rank == 0 and world == 1 and not args.no_extras and args.cfg == 2 and args.mode == "bf16":
        try:
            from src.runtime.graphed import GraphedTrainStep
            m3, n3, o3 = build("bf16", 3)
            g3 = torch.Generator(device=dev).manual_seed(4321)
            b3 = (torch.rand(32, 3, 64, 64, device=dev, generator=g3) * 2 - 1, None)
            o3.device_state = True
            for i in range(3):
                l3 = m3.training_step(b3, i); l3.backward(); o3.step()
            gstep3 = GraphedTrainStep(m3, o3, b3, warmup=0)
            ms3 = timed_leg(lambda i: gstep3(b3), 30, 25, torch.cuda.synchronize, 1, dev)
            fin3 = bool(torch.isfinite(n3.flat_params).all())
            cfg3 = {"value": round(32 / ms3 * 1e3, 1) if fin3 else None, "unit": "images/s", "ms_per_step": round(ms3, 3), "per_gpu_batch": 32,
                    "train_tflops": round(32 / ms3 * 26.278, 1), "step_launch": "hipGraph replay", "state_finite": fin3}
            del gstep3, m3, n3, o3
        except Exception as exc:                            # noqa: BLE001  (an extra: reported, never fatal)
            cfg3 = {"error": f"{type(exc).__name__}: {exc}"[:300]}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.no_extras and args.cfg == 2:
        cpu = cpu_baseline()

    if rank == 0:
        def best_named(sto):
            """(hbm_frac, unit_us, which) of the best unit of the named kernel for one activation storage"""
            ent = ((roof or {}).get("named_kernel") or {}).get(sto + "_storage") or {}
            cands = [(v["hbm_frac"], v["unit_us"], k) for k, v in ent.items() if isinstance(v, dict) and "hbm_frac" in v and k != "two_pass"]
            return max(cands) if cands else (None, None, None)
        nk32, nk16 = best_named("fp32"), best_named("bf16")
        nk3 = ((roof or {}).get("named_kernel_cfg3_level0") or {})
        nk3_32 = max([v["hbm_frac"] for v in (nk3.get("fp32_storage") or {}).values() if isinstance(v, dict) and "hbm_frac" in v] or [None], key=lambda v: v or 0)
        workload = (("DDPM CIFAR-10 32x32, UNet 128/1-2-4, T=1000 (BASELINE configs[1]): train step B=128/GPU + hipGraph denoise step B=64")
                    if args.cfg == 2 else
                    ("DDPM CelebA 64x64, UNet 64/1-2-4-8, T=1000 (BASELINE configs[2], 32/GPU at 8 GPUs): train step + denoise step B=64"))
        # The LAST line is the record: every number of BASELINE's metric is a SCALAR key of `config` or `roofline` (the driver keeps scalars
        # and truncates strings at ~120 characters); the per-kernel tables, the named-kernel legs, the fp32-mode and data-parallel legs
        # go out one line earlier ("detail") and to gpurun_out/bench_detail.json when that directory exists.
        detail = {"detail": "bench.py per-kernel tables and extra legs (the record is the next line)",
                  "all_kernels": (roof or {}).pop("all_kernels", None), "named_kernel": (roof or {}).pop("named_kernel", None),
                  "named_kernel_cfg3_level0": (roof or {}).pop("named_kernel_cfg3_level0", None),
                  "fp32_mode": fp32_mode, "dp_path_n1": dp_path, "cfg3_b32": cfg3, "comm": comm, "rank_ms_per_step": spread,
                  "cpu_baseline_legs": (cpu or {}).pop("legs", None),
                  "traffic_note": (roof or {}).pop("traffic_note", None)}
        out = {
            "metric": "ddpm_cifar10_32x32_train_images_per_sec" if args.cfg == 2 else "ddpm_celeba_64x64_train_images_per_sec",
            "value": round(images_per_s, 1), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.mode,
            "data": "synthetic",
            "config": {"workload": workload,
                       "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"dp{world}",
                       "step_launch": ("hipGraph replay" if use_graph else "eager") + (f" [{graph_note}]"[:60] if graph_note else ""),
                       # BASELINE's metric has two halves; the second one (hipGraph-replayed reverse-diffusion step at B = 64):
                       "denoise_steps_per_sec": round(denoise_steps_per_s, 2), "denoise_batch": 64,
                       "denoise_tflops": round(denoise_steps_per_s * 64 * FWD_GF / 1e3, 1),
                       "train_tflops": round(images_per_s * TRAIN_GF / 1e3, 1),
                       "dp_path_n1_images_per_sec": (dp_path or {}).get("value"),
                       "fp32_mode_images_per_sec": (fp32_mode or {}).get("value"),
                       "cfg3_b32_images_per_sec": (cfg3 or {}).get("value"),
                       "named_kernel_hbm_frac_fp32": nk32[0], "named_kernel_unit_us_fp32": nk32[1],
                       "named_kernel_hbm_frac_bf16": nk16[0], "named_kernel_unit_us_bf16": nk16[1],
                       "named_kernel_cfg3_level0_hbm_frac_fp32": nk3_32,
                       "sclk_mhz_under_mfma": sclk,
                       "rccl_ranks": rccl_ranks,
                       "allreduce_ms_exposed": (comm or {}).get("allreduce_ms_exposed"),
                       "comm_exposed_ms_by_subtraction": ((comm or {}).get("ways_ms") or {}).get("exposed_comm_ms"),
                       "eager_overlap_ms_per_step": ((comm or {}).get("ways_ms") or {}).get("eager_overlap"),
                       "graph_segments": (comm or {}).get("graph_segments"),
                       "rank_ms_min": spread.get("min"), "rank_ms_max": spread.get("max"),
                       "final_loss": round(final_loss, 5), "state_finite": state_finite,
                       "matmul": "bf16 MFMA, fp32 accumulate" if args.mode == "bf16" else "fp32 MFMA"},
            "roofline": roof, "cpu_baseline": cpu,
        }
        # (RCCL writes its version banner through C stdio, which is flushed at exit -- i.e. AFTER a line printed from Python: flush it
        #  first so that the JSON line is the last line of the output)
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:       # noqa: BLE001
            pass
        print(json.dumps(detail), flush=True)
        try:            # the full report in one object (the layout tools/update_design_table.py and update_profiles_readme.py read)
            if os.path.isdir(os.path.join(ROOT, "gpurun_out")):
                full = dict(out)
                full["roofline"] = dict(roof or {}, all_kernels=detail["all_kernels"], named_kernel=detail["named_kernel"],
                                        named_kernel_cfg3_level0=detail["named_kernel_cfg3_level0"], traffic_note=detail["traffic_note"])
                full["cpu_baseline"] = dict(cpu or {}, legs=detail["cpu_baseline_legs"]) if cpu else None
                full.update({"fp32_mode": fp32_mode, "dp_path_n1": dp_path, "cfg3_b32": cfg3, "comm": comm, "rank_ms_per_step": spread,
                             "denoise_steps_per_sec": round(denoise_steps_per_s, 2), "denoise_tflops": out["config"]["denoise_tflops"],
                             "train_tflops": out["config"]["train_tflops"]})
                with open(os.path.join(ROOT, "gpurun_out", "bench_detail.json"), "w") as f:
                    json.dump(full, f, indent=1)
        except OSError:
            pass
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
