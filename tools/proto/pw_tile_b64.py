"""conv_pw at the sampler's batch (B = 64): 64- vs 128-pixel tiles on the shapes whose 128-pixel grid is about one workgroup per CU."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-generation-models_amd")]
import torch
from src.ops import functional as K
lib = K.load_library(); K.PW_MIN_TILES = 0


def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for B in (64, 128):
    for H, Ci, Co in ((32, 128, 128), (16, 256, 256), (16, 512, 256), (16, 256, 128), (8, 512, 512), (8, 1024, 512)):
        x = torch.randn(B, H, H, Ci, device="cuda").bfloat16()
        w = torch.randn(3, 3, Ci, Co, device="cuda") * 0.05
        table, nent, tiles = K.pack_table([(0, 9, Ci, Co)], "cuda")
        wd, wf, wdq, wfq = (torch.zeros(w.numel(), device="cuda", dtype=torch.bfloat16) for _ in range(4))
        K.pack_weights_bf16(table, nent, tiles, w.reshape(-1), wd, wf, wdq, wfq)
        y = torch.empty(B, H, H, Co, device="cuda", dtype=torch.bfloat16)
        res = []
        for tile in (0, 64, 128):
            lib.mi_debug_conv_pw_tile(tile)
            try:
                t = timeit(lambda: K.conv3x3_bf16w(x, wf, K=Ci, Nc=Co, flip=False, out=y, wq=wfq))
            except Exception as e:
                t = float("nan")
            res.append(t)
        lib.mi_debug_conv_pw_tile(0)
        t128 = B * H * H // 128 * ((Co + 127) // 128)
        print(f"B{B} {H}x{H} {Ci}->{Co}: 128-px workgroups {t128:5d} | auto {res[0]:6.1f} us  tile64 {res[1]:6.1f}  tile128 {res[2]:6.1f}", flush=True)
