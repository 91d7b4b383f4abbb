"""Node types of the captured DDPM training step and of the captured denoise iteration (hipGraphGetNodes / hipGraphNodeGetType): how many
memset / memcpy nodes ride in them beside the kernel nodes."""
import ctypes, os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-generation-models_amd")]
import torch
from src.models.ddpm import DDPM
hip = ctypes.CDLL("libamdhip64.so")
NAMES = {0: "kernel", 1: "memcpy", 2: "memset", 3: "host", 4: "graph", 5: "empty", 6: "waitEvent", 7: "eventRecord"}


def node_types(g):
    raw = g.raw_cuda_graph()
    n = ctypes.c_size_t(0)
    assert hip.hipGraphGetNodes(ctypes.c_void_p(raw), None, ctypes.byref(n)) == 0
    arr = (ctypes.c_void_p * n.value)()
    assert hip.hipGraphGetNodes(ctypes.c_void_p(raw), arr, ctypes.byref(n)) == 0
    c = collections.Counter()
    for i in range(n.value):
        t = ctypes.c_int(-1)
        hip.hipGraphNodeGetType(ctypes.c_void_p(arr[i]), ctypes.byref(t))
        c[NAMES.get(t.value, str(t.value))] += 1
    return dict(c)


B = 128
torch.manual_seed(0)
m = DDPM({"width": 32, "height": 32, "channels": 3, "transforms": {"normalize": True}}, hidden_dim=128, dim_mults=(1, 2, 4), lr=1e-4, b1=0.9, b2=0.999).to("cuda")
m.denoising_model.compute_mode = "bf16"; m.train(); m.log = lambda *a, **k: None
opt = m.configure_optimizers(); opt.device_state = True
x = torch.rand(B, 3, 32, 32, device="cuda") * 2 - 1
for i in range(3):
    l = m.training_step((x, None), i); l.backward(); opt.step()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    g = torch.cuda.CUDAGraph(keep_graph=True)
    with torch.cuda.graph(g, stream=s):
        l = m.training_step((x, None), 0); l.backward(); opt.step()
print("training step:", node_types(g))
from src.runtime.sampler import GraphSampler
m.eval()
gs = GraphSampler(m.diffusion_model, (64, 3, 32, 32))
gs.refresh()
with torch.cuda.stream(s):
    gs.t.fill_(1); gs.set_image(torch.zeros_like(gs.x)); gs._iteration()
    g2 = torch.cuda.CUDAGraph(keep_graph=True)
    with torch.cuda.graph(g2, stream=s):
        gs._iteration()
print("denoise iteration:", node_types(g2))
