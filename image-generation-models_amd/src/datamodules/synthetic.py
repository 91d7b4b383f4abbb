"""Synthetic image data (uniform noise images in [-1,1] after normalisation) for smoke runs and
benchmarks on boxes without datasets or network."""
import numpy as np

from .base import ArrayImageDataset, BaseDatamodule


class SyntheticDataModule(BaseDatamodule):
    def __init__(self, width=32, height=32, channels=3, batch_size: int = 128, num_workers: int = 0,
                 train_size: int = 1024, val_size: int = 128, transforms=None, seed: int = 0, **kargs):
        super().__init__(width, height, channels, batch_size, num_workers, kargs.get("device_resident", "auto"))
        self.train_size, self.val_size, self.transforms, self.seed = train_size, val_size, transforms, seed

    def setup(self, stage=None):
        rng = np.random.default_rng(self.seed)

        def make(n):
            x = rng.integers(0, 256, size=(n, self.height, self.width, self.channels), dtype=np.uint8)
            return ArrayImageDataset(x, np.zeros(n, dtype=np.int64), self.transforms)
        self.train_data, self.val_data = make(self.train_size), make(self.val_size)
