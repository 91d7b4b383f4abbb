import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-generation-models_amd")]
import torch
from src.ops import functional as K
for us in (300, 1000, 3000, 10000, 3000, 300):
    print(us, [K.clock_probe("cuda", usec=us) for _ in range(4)])
