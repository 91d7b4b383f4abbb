"""`SampleImagesCallback`: on validation batch 0 write real / sample / "diffusion" image grids to
the logger and `results/<epoch>.jpg` (reference: src/callbacks/visualization.py:13-38,141-147).
The grid is torchvision.utils.make_grid's layout restated (nrow=8, padding=2, pad_value=1,
normalize to value_range (-1,1) -> [0,1]); torchvision itself is not needed."""
from pathlib import Path

import numpy as np
import torch

try:
    from pytorch_lightning import Callback
except ImportError:
    from src.runtime.lightning_lite import Callback


def make_grid(imgs: torch.Tensor, nrow: int = 8, padding: int = 2, pad_value: float = 1.0, normalize: bool = False,
              value_range=None) -> torch.Tensor:
    imgs = imgs.detach().float().cpu()
    if imgs.dim() == 3:
        imgs = imgs[None]
    if imgs.shape[1] == 1:
        imgs = imgs.repeat(1, 3, 1, 1)
    if normalize:
        lo, hi = value_range if value_range is not None else (float(imgs.min()), float(imgs.max()))
        imgs = ((imgs.clamp(lo, hi) - lo) / max(hi - lo, 1e-5))
    n, c, h, w = imgs.shape
    if n == 1:
        return imgs[0]
    xmaps = min(nrow, n)
    ymaps = (n + xmaps - 1) // xmaps
    H, W = h + padding, w + padding
    grid = torch.full((c, H * ymaps + padding, W * xmaps + padding), float(pad_value))
    k = 0
    for y in range(ymaps):
        for x in range(xmaps):
            if k >= n:
                break
            grid[:, y * H + padding:y * H + padding + h, x * W + padding:x * W + padding + w] = imgs[k]
            k += 1
    return grid


def get_grid_images(imgs, model, nimgs=64, nrow=8):
    if model.input_normalize:
        return make_grid(imgs[:nimgs], nrow=nrow, normalize=True, value_range=(-1, 1), pad_value=1)
    return make_grid(imgs[:nimgs], normalize=False, nrow=nrow, pad_value=1)


def save_image(grid: torch.Tensor, path):
    from PIL import Image
    arr = (grid.clamp(0, 1).numpy().transpose(1, 2, 0) * 255 + 0.5).astype(np.uint8)
    Image.fromarray(arr).save(str(path))


class SampleImagesCallback(Callback):
    def __init__(self, batch_size=64, every_n_epochs=1):
        self.batch_size = batch_size
        self.every_n_epochs = every_n_epochs

    def on_validation_batch_end(self, trainer, pl_module, outputs, batch, batch_idx, dataloader_idx=0):
        if trainer.current_epoch % self.every_n_epochs != 0 or batch_idx != 0:
            return
        if not getattr(trainer, "is_global_zero", True):
            return
        out_dir = Path("results")
        out_dir.mkdir(parents=True, exist_ok=True)
        exp = getattr(getattr(trainer, "logger", None), "experiment", None)

        def emit(tag, imgs, save=False):
            if imgs is None:
                return
            grid = get_grid_images(imgs, pl_module)
            if exp is not None:
                exp.add_image(f"images/{tag}", grid, global_step=trainer.current_epoch)
            if save:
                save_image(grid, out_dir / f"{trainer.current_epoch}.jpg")

        emit("real", outputs.real_image)
        emit("recon", outputs.recon_image)
        emit("sample", outputs.fake_image, save=True)
        for key, val in outputs.others.items():
            emit(key, val)
