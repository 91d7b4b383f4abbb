"""Do an MFMA-bound conv_pw stream and an HBM-bound streaming kernel (or the batched weight gradient) overlap when they sit
on two HIP streams?  Serial time vs concurrent time of the same launches (eager, and captured as one hipGraph with a fork)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-generation-models_amd")]
import torch
from src.ops import functional as K

B = 128
lib = K.load_library(); K.PW_MIN_TILES = 0


def mk(H, Ci, Co):
    x = torch.randn(B, H, H, Ci, device="cuda").bfloat16()
    w = torch.randn(3, 3, Ci, Co, device="cuda") * 0.05
    table, nent, tiles = K.pack_table([(0, 9, Ci, Co)], "cuda")
    wd, wf, wdq, wfq = (torch.zeros(w.numel(), device="cuda", dtype=torch.bfloat16) for _ in range(4))
    K.pack_weights_bf16(table, nent, tiles, w.reshape(-1), wd, wf, wdq, wfq)
    y = torch.empty(B, H, H, Co, device="cuda", dtype=torch.bfloat16)
    return lambda: K.conv3x3_bf16w(x, wf, K=Ci, Nc=Co, flip=False, out=y, wq=wfq)


convs = [mk(32, 128, 128), mk(16, 256, 256), mk(8, 512, 512)]
big = torch.randn(64 << 20, device="cuda")          # 256 MB fp32: a streaming read-modify-write of 512 MB


def conv_leg(n=10):
    for _ in range(n):
        for c in convs:
            c()


def stream_leg(n=3):
    for _ in range(n):
        big.mul_(1.0001)


# the batched 3x3 weight gradient of eight level-0 layers
X = [torch.randn(B, 32, 32, 128, device="cuda").bfloat16() for _ in range(8)]
DY = [torch.randn(B, 32, 32, 128, device="cuda").bfloat16() for _ in range(8)]
DW = [torch.zeros(3, 3, 128, 128, device="cuda") for _ in range(8)]


def wgrad_leg():
    wq = K.WgradQueue(group=8)
    for x, dy, dw in zip(X, DY, DW):
        wq.push(x, dy, dw, Ci=128, Cj=128, hw=(32, 32), mode=K.MODE_BF16)
    wq.flush()


side = torch.cuda.Stream()


def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def both(other):
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        other()
    conv_leg()
    main.wait_stream(side)


for name, other in (("streaming 3 x 512 MB", stream_leg), ("batched wgrad 8 x level 0", wgrad_leg)):
    a, b = timed(conv_leg), timed(other)
    c = timed(lambda: both(other))
    print(f"eager : convs {a:7.1f} us, {name} {b:7.1f} us, serial {a + b:7.1f}, two streams {c:7.1f} us", flush=True)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        g1 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1, stream=s):
            conv_leg(); other()
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2, stream=s):
            both(other)
    torch.cuda.current_stream().wait_stream(s)
    print(f"graph : serial {timed(g1.replay):7.1f} us, forked {timed(g2.replay):7.1f} us", flush=True)
