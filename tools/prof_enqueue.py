import os, sys, cProfile, pstats
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-generation-models_amd")]
import torch
from src.models.ddpm import DDPM
B = 8
torch.manual_seed(0)
m = DDPM({"width": 32, "height": 32, "channels": 3, "transforms": {"normalize": True}}, hidden_dim=128, dim_mults=(1, 2, 4), lr=1e-4, b1=0.9, b2=0.999).to("cuda")
m.denoising_model.compute_mode = "bf16"; m.train()
opt = m.configure_optimizers()
x = torch.rand(B, 3, 32, 32, device="cuda") * 2 - 1
for i in range(5):
    loss = m.training_step((x, None), i); loss.backward(); opt.step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for i in range(20):
    loss = m.training_step((x, None), i); loss.backward(); opt.step()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(22)
