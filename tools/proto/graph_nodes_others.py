"""Node types of the captured VAE / VQ-VAE training steps (src/runtime/graphed.py::node_types)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-generation-models_amd")]
import torch
G = importlib.import_module("src.runtime.graphed")
OPT = importlib.import_module("src.runtime.optim")
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_vae_gpu as TV
m = TV._model(32, 128, seed=3).cuda().train(); m.log = lambda *a, **k: None
opt = OPT.FlatAdam(m.flat_nets(), lr=1e-3, betas=(0.9, 0.999), device_state=True)
x = torch.rand(64, 1, 28, 28, device="cuda") * 2 - 1
step = G.GraphedTrainStep(m, opt, (x, None))
print("VAE step:", G.node_types(step.graph))
try:
    import test_vqvae_gpu as TQ
    src = open(os.path.join(ROOT, "tests", "test_vqvae_gpu.py")).read()
    print("vqvae test helpers:", [l for l in src.splitlines() if l.startswith("def _")][:6])
except Exception as e:
    print("vqvae import:", e)
