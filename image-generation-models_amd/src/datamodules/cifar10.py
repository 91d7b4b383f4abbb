"""CIFAR-10 from the standard `cifar-10-batches-py` pickles (what torchvision's CIFAR10 reads,
reference: src/datamodules/cifar10.py:20-28).  No network here, so nothing is downloaded:
a missing dataset is an error that says where the files are expected."""
import os
import pickle

import numpy as np

from .base import ArrayImageDataset, BaseDatamodule


def _read_batches(root, names):
    xs, ys = [], []
    for n in names:
        path = os.path.join(root, "cifar-10-batches-py", n)
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} not found: unpack cifar-10-python.tar.gz under data_dir (no download on this box)")
        with open(path, "rb") as f:
            d = pickle.load(f, encoding="latin1")
        xs.append(np.asarray(d["data"], dtype=np.uint8).reshape(-1, 3, 32, 32).transpose(0, 2, 3, 1))
        ys.append(np.asarray(d.get("labels", d.get("fine_labels")), dtype=np.int64))
    return np.concatenate(xs), np.concatenate(ys)


class CIFAR10DataModule(BaseDatamodule):
    def __init__(self, data_dir: str = "./data", width=64, height=64, channels=3, batch_size: int = 64,
                 num_workers: int = 8, transforms=None, **kargs):
        super().__init__(width, height, channels, batch_size, num_workers, kargs.get("device_resident", "auto"))
        self.data_dir, self.transforms = data_dir, transforms

    def prepare_data(self):
        _read_batches(self.data_dir, ["test_batch"])            # existence check only

    def setup(self, stage=None):
        xtr, ytr = _read_batches(self.data_dir, [f"data_batch_{i}" for i in range(1, 6)])
        xte, yte = _read_batches(self.data_dir, ["test_batch"])
        self.train_data = ArrayImageDataset(xtr, ytr, self.transforms)
        self.val_data = ArrayImageDataset(xte, yte, self.transforms)
