#!/usr/bin/env python3
"""How does the training step behave while another stream holds part of the chip (what an overlapping RCCL all-reduce does at N > 1)?
Runs training steps while `H` spinning workgroups occupy a side stream, for H in a list, with the default one-workgroup-per-CU
weight-gradient plan and with smaller plans (MI_W3_BLOCKS).   python tools/cu_hog.py"""
import ctypes as C
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(hog):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "image-generation-models_amd")]
    import torch
    from src.models.ddpm import DDPM
    from src.ops.lib import load_library
    lib = load_library()
    torch.manual_seed(0)
    m = DDPM({"width": 32, "height": 32, "channels": 3, "transforms": {"normalize": True}}, hidden_dim=128, dim_mults=(1, 2, 4), lr=1e-4, b1=0.9, b2=0.999).to("cuda")
    m.denoising_model.compute_mode = "bf16"; m.train()
    opt = m.configure_optimizers()
    imgs = torch.rand(128, 3, 32, 32, device="cuda") * 2 - 1
    def step(i):
        loss = m.training_step((imgs, None), i); loss.backward(); opt.step()
    for i in range(40):
        step(i)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    per = 1 << 20                                            # 4 MB read + 4 MB written per workgroup and pass
    sink = torch.zeros(max(hog, 1) * 2 * per, device="cuda")
    mode = int(os.environ.get("HOG_MODE", "1"))
    n = 40
    if hog:
        lib.mi_debug_spin(hog, 600_000, C.c_void_p(sink.data_ptr()), per, mode, C.c_void_p(side.cuda_stream))     # 0.6 s: covers the timed steps
        time.sleep(0.02)
    t0 = time.perf_counter()
    for i in range(n):
        step(i)
    torch.cuda.current_stream().synchronize()
    el = time.perf_counter() - t0
    print(f"hog {hog:3d} workgroups mode {mode} MI_W3_BLOCKS={os.environ.get('MI_W3_BLOCKS', '256'):>4s}: {el / n * 1e3:7.3f} ms/step", flush=True)
    torch.cuda.synchronize()


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(int(sys.argv[1]))
    else:
        blocks = os.environ.get("HOG_W3_BLOCKS", "256").split(",")
        # 0 = ALU spin (worst case for issue slots), 1 = streaming copy at full tilt for the whole step, 2 = the same copy with a
        # gradient all-reduce's duty cycle (0.5 ms of every 5 ms; resident but asleep in between)
        for mode in os.environ.get("HOG_MODES", "2,1,0").split(","):
            for b in blocks:
                for hog in (0, 8, 32, 64):
                    subprocess.run([sys.executable, __file__, str(hog)], env={**os.environ, "HOG_MODE": mode, "MI_W3_BLOCKS": b}, check=False)
