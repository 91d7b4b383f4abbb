"""MNIST from the raw idx files (`MNIST/raw/*-ubyte[.gz]`, the layout torchvision writes;
reference: src/datamodules/mnist.py)."""
import gzip
import os
import struct

import numpy as np

from .base import ArrayImageDataset, BaseDatamodule


def _idx(path):
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rb") as f:
        _, _, dtype, ndim = struct.unpack(">BBBB", f.read(4))
        shape = struct.unpack(">" + "I" * ndim, f.read(4 * ndim))
        return np.frombuffer(f.read(), dtype=np.uint8).reshape(shape)


def _find(root, stem):
    for d in (os.path.join(root, "MNIST", "raw"), root):
        for ext in ("", ".gz"):
            p = os.path.join(d, stem + ext)
            if os.path.exists(p):
                return p
    raise FileNotFoundError(f"{stem} not found under {root}/MNIST/raw (no download on this box)")


class MNISTDataModule(BaseDatamodule):
    def __init__(self, data_dir: str = "./data", width=28, height=28, channels=1, batch_size: int = 64,
                 num_workers: int = 8, transforms=None, **kargs):
        super().__init__(width, height, channels, batch_size, num_workers, kargs.get("device_resident", "auto"))
        self.data_dir, self.transforms = data_dir, transforms

    def prepare_data(self):
        _find(self.data_dir, "t10k-images-idx3-ubyte")

    def setup(self, stage=None):
        def load(kind):
            x = _idx(_find(self.data_dir, f"{kind}-images-idx3-ubyte"))[:, :, :, None]
            y = _idx(_find(self.data_dir, f"{kind}-labels-idx1-ubyte")).astype(np.int64)
            return ArrayImageDataset(x, y, self.transforms)
        self.train_data, self.val_data = load("train"), load("t10k")
