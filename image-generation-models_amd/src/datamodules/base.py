"""Data modules without torchvision: raw dataset files -> uint8 arrays -> `(x/255 - 0.5)/0.5`
float tensors in NCHW (reference pipeline: ToTensor + Normalize(0.5, 0.5),
src/datamodules/base.py:37-71), served by torch DataLoaders with the reference's settings
(shuffle on train, fork workers, no pin_memory / drop_last, base.py:14-27).  Under data-parallel
training each rank reads a disjoint, equally sized shard per epoch (DistributedSampler semantics)."""
from __future__ import annotations

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset, Sampler

try:
    from pytorch_lightning import LightningDataModule
except ImportError:
    from src.runtime.lightning_lite import LightningDataModule


def _cfg_get(cfg, key, default=None):
    if cfg is None:
        return default
    return cfg.get(key, default) if isinstance(cfg, dict) else getattr(cfg, key, default)


class ArrayImageDataset(Dataset):
    """uint8 images [N,H,W,C] (+ int labels) with the reference transform chain applied on access."""

    def __init__(self, images: np.ndarray, labels: np.ndarray, transforms=None):
        self.images, self.labels = images, labels
        self.normalize = bool(_cfg_get(transforms, "normalize", False))
        self.flip = _cfg_get(transforms, "flip") is not None
        self.resize = _cfg_get(transforms, "resize")

    def __len__(self):
        return len(self.images)

    def __getitem__(self, i):
        img = self.images[i]
        if self.resize is not None:
            from PIL import Image
            size = (int(_cfg_get(self.resize, "width")), int(_cfg_get(self.resize, "height")))
            method = {"nearest": Image.NEAREST, "bilinear": Image.BILINEAR}.get(_cfg_get(self.resize, "method"), Image.BICUBIC)
            img = np.asarray(Image.fromarray(img.squeeze()).resize(size, method))
            if img.ndim == 2:
                img = img[:, :, None]
        x = torch.from_numpy(np.ascontiguousarray(img)).permute(2, 0, 1).float().div_(255.0)     # ToTensor
        if self.flip and torch.rand(()) < 0.5:
            x = x.flip(-1)
        if self.normalize:
            x = (x - 0.5) / 0.5
        return x, int(self.labels[i])


class ShardSampler(Sampler):
    """DistributedSampler semantics: per-epoch seeded shuffle, padded to equal shards."""

    def __init__(self, n, rank=0, world=1, shuffle=True, seed=0):
        self.n, self.rank, self.world, self.shuffle, self.seed, self.epoch = n, rank, world, shuffle, seed, 0
        self.per = (n + world - 1) // world

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __len__(self):
        return self.per

    def __iter__(self):
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed + self.epoch)
            idx = torch.randperm(self.n, generator=g).tolist()
        else:
            idx = list(range(self.n))
        idx += idx[: self.per * self.world - self.n]
        return iter(idx[self.rank::self.world])


class BaseDatamodule(LightningDataModule):
    def __init__(self, width, height, channels, batch_size, num_workers):
        super().__init__()
        self.width, self.height, self.channels = width, height, channels
        self.batch_size, self.num_workers = batch_size, num_workers
        self._rank, self._world = 0, 1
        self._train_sampler = None

    def set_shard(self, rank, world):
        self._rank, self._world = rank, world

    def set_epoch(self, epoch):
        if self._train_sampler is not None:
            self._train_sampler.set_epoch(epoch)

    def _loader(self, data, shuffle):
        kw = dict(batch_size=self.batch_size, num_workers=self.num_workers)
        if self.num_workers > 0:
            kw["multiprocessing_context"] = "fork"
        if self._world > 1:
            sampler = ShardSampler(len(data), self._rank, self._world, shuffle=shuffle)
            if shuffle:
                self._train_sampler = sampler
            return DataLoader(data, sampler=sampler, **kw)
        return DataLoader(data, shuffle=shuffle, **kw)

    def train_dataloader(self):
        return self._loader(self.train_data, True)

    def val_dataloader(self):
        return self._loader(self.val_data, False)
