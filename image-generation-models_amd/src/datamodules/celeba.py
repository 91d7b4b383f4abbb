"""CelebA from a folder of aligned jpgs (`celeba/img_align_celeba/*.jpg`), resized to
(height, width) with BICUBIC and no crop, as configs/datamodule/celeba.yaml asks
(reference: src/datamodules/celeba.py, base.py:43-49)."""
import os

import numpy as np
from torch.utils.data import Dataset

from .base import ArrayImageDataset, BaseDatamodule


class _JpgFolder(Dataset):
    def __init__(self, files, transforms):
        self.files = files
        self._tf = ArrayImageDataset(np.zeros((0,)), np.zeros((0,)), transforms)

    def __len__(self):
        return len(self.files)

    def load_uint8(self, i) -> np.ndarray:
        """decoded RGB jpg, resized as the config asks (BICUBIC, no crop): uint8 [height, width, 3]"""
        from PIL import Image
        return self._tf.resized(np.asarray(Image.open(self.files[i]).convert("RGB")))

    def __getitem__(self, i):
        from PIL import Image
        img = np.asarray(Image.open(self.files[i]).convert("RGB"))
        self._tf.images, self._tf.labels = [img], [0]
        return self._tf[0]


class CelebADataModule(BaseDatamodule):
    def __init__(self, data_dir: str = "./data", width=64, height=64, channels=3, batch_size: int = 64,
                 num_workers: int = 8, transforms=None, **kargs):
        super().__init__(width, height, channels, batch_size, num_workers, kargs.get("device_resident", "auto"))
        self.data_dir, self.transforms = data_dir, transforms

    def _files(self):
        root = os.path.join(self.data_dir, "celeba", "img_align_celeba")
        if not os.path.isdir(root):
            raise FileNotFoundError(f"{root} not found (no download on this box)")
        return sorted(os.path.join(root, f) for f in os.listdir(root) if f.endswith(".jpg"))

    def prepare_data(self):
        self._files()

    def setup(self, stage=None):
        files = self._files()
        n_train = int(len(files) * 0.9)             # torchvision's split files are not required
        self.train_data = _JpgFolder(files[:n_train], self.transforms)
        self.val_data = _JpgFolder(files[n_train:], self.transforms)
