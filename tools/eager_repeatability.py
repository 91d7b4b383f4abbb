import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-generation-models_amd")]
import torch
from src.models.ddpm import DDPM
DEV="cuda"
dm = {"width": 16, "height": 16, "channels": 3, "transforms": {"normalize": True}}
def run(mode):
    torch.manual_seed(0)
    m = DDPM(dm, hidden_dim=32, dim_mults=(1, 2), timesteps=1000, lr=2e-3, b1=0.9, b2=0.999).to(DEV).train()
    m.denoising_model.compute_mode = mode; m.log = lambda *a, **k: None
    o = m.configure_optimizers()
    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.rand(16, 3, 16, 16, device=DEV, generator=g) * 2 - 1
    torch.manual_seed(11)
    out=[]
    for i in range(24):
        l = m.training_step((x, None), i); l.backward(); o.step(); out.append(float(l))
    return out
for mode in ("fp32","bf16"):
    a=run(mode); b=run(mode); c=run(mode)
    print(mode, "eager vs eager max diff", max(abs(x-y) for x,y in zip(a,b)), max(abs(x-y) for x,y in zip(a,c)), [round(abs(x-y),5) for x,y in zip(a,b)][:24:4])
