"""Data path on the host: the MNIST idx reader and the CelebA jpg reader against hand-made files, the
reference transform chain (reference src/datamodules/base.py:37-71, configs/datamodule/celeba.yaml:14-19: Resize([H, W], BICUBIC)
with NO crop -> ToTensor -> Normalize(0.5, 0.5)), and the uint8 materialisation that feeds the device-resident loader."""
import gzip
import os
import struct

import numpy as np
import pytest
import torch


def _write_idx(path, arr: np.ndarray, gz=False):
    """idx format: 2 zero bytes, dtype code 0x08 (uint8), ndim, then big-endian uint32 dims, then the bytes."""
    blob = struct.pack(">BBBB", 0, 0, 8, arr.ndim) + struct.pack(">" + "I" * arr.ndim, *arr.shape) + arr.astype(np.uint8).tobytes()
    (gzip.open if gz else open)(path, "wb").write(blob)


@pytest.mark.parametrize("gz,nested", [(False, True), (True, True), (True, False)])
def test_mnist_idx_reader(tmp_path, gz, nested):
    from src.datamodules.mnist import MNISTDataModule
    d = tmp_path / "MNIST" / "raw" if nested else tmp_path
    d.mkdir(parents=True, exist_ok=True)
    rng = np.random.default_rng(1)
    ext = ".gz" if gz else ""
    xtr, ytr = rng.integers(0, 256, (12, 28, 28), dtype=np.uint8), rng.integers(0, 10, 12, dtype=np.uint8)
    xte, yte = rng.integers(0, 256, (5, 28, 28), dtype=np.uint8), rng.integers(0, 10, 5, dtype=np.uint8)
    xte[0, 0, :3] = (0, 255, 128)                                      # the three values whose images are -1, 1, 0.00392...
    _write_idx(str(d / f"train-images-idx3-ubyte{ext}"), xtr, gz); _write_idx(str(d / f"train-labels-idx1-ubyte{ext}"), ytr, gz)
    _write_idx(str(d / f"t10k-images-idx3-ubyte{ext}"), xte, gz); _write_idx(str(d / f"t10k-labels-idx1-ubyte{ext}"), yte, gz)
    dm = MNISTDataModule(str(tmp_path), 28, 28, 1, batch_size=4, num_workers=0, transforms={"convert": True, "normalize": True})
    dm.prepare_data(); dm.setup()
    assert len(dm.train_data) == 12 and len(dm.val_data) == 5
    batches = list(dm.val_dataloader())
    assert [b[0].shape[0] for b in batches] == [4, 1]                   # ragged last batch, no drop_last (reference base.py:23-27)
    x, y = batches[0]
    assert x.shape == (4, 1, 28, 28) and x.dtype == torch.float32 and y.dtype == torch.int64
    assert torch.equal(y, torch.from_numpy(yte[:4].astype(np.int64)))
    assert torch.equal(x[:, 0], (torch.from_numpy(xte[:4]).float() / 255 - 0.5) / 0.5)       # ToTensor then Normalize(0.5, 0.5)
    assert x[0, 0, 0, 0] == -1.0 and x[0, 0, 0, 1] == 1.0 and abs(float(x[0, 0, 0, 2]) - (128 / 255 - 0.5) / 0.5) < 1e-7
    # without `normalize` the chain stops at ToTensor: [0, 1]
    dm2 = MNISTDataModule(str(tmp_path), 28, 28, 1, batch_size=4, num_workers=0, transforms={"convert": True, "normalize": False})
    dm2.setup()
    x2, _ = next(iter(dm2.val_dataloader()))
    assert torch.equal(x2[:, 0], torch.from_numpy(xte[:4]).float() / 255)


def test_mnist_missing_files_fail_loudly(tmp_path):
    from src.datamodules.mnist import MNISTDataModule
    with pytest.raises(FileNotFoundError):
        MNISTDataModule(str(tmp_path), 28, 28, 1, batch_size=4, num_workers=0).prepare_data()


def _celeba_folder(tmp_path, n=10, w=178, h=218):
    """jpgs at CelebA's aligned size (178 x 218, portrait): left half red, right half blue, a green band across the top rows whose
    height depends on the file, so that crop / transposition / file order mistakes show."""
    from PIL import Image
    root = tmp_path / "celeba" / "img_align_celeba"
    root.mkdir(parents=True)
    for i in range(n):
        img = np.zeros((h, w, 3), dtype=np.uint8)
        img[:, : w // 2, 0] = 255
        img[:, w // 2:, 2] = 255
        img[: 20 + 10 * i, :, 1] = 200
        Image.fromarray(img).save(str(root / f"{i + 1:06d}.jpg"), quality=95)
    (root / "notes.txt").write_text("not an image")
    return root


def test_celeba_jpg_reader_resize_no_crop(tmp_path):
    from PIL import Image
    from src.datamodules.celeba import CelebADataModule
    root = _celeba_folder(tmp_path)
    tf = {"convert": True, "normalize": True, "resize": {"height": 64, "width": 64}}
    dm = CelebADataModule(str(tmp_path), 64, 64, 3, batch_size=4, num_workers=0, transforms=tf)
    dm.prepare_data(); dm.setup()
    assert len(dm.train_data) == 9 and len(dm.val_data) == 1              # 90 / 10 split over the sorted file list
    x, _ = next(iter(dm.val_dataloader()))
    assert x.shape == (1, 3, 64, 64) and x.dtype == torch.float32
    # the expected tensor, step by step as torchvision does it for a PIL image: Resize -> PIL.resize((W, H), BICUBIC)
    pil = Image.open(str(root / "000010.jpg")).convert("RGB").resize((64, 64), Image.BICUBIC)
    want = (torch.from_numpy(np.asarray(pil)).permute(2, 0, 1).float() / 255 - 0.5) / 0.5
    assert torch.equal(x[0], want)
    # no crop and no transposition: the WHOLE width is kept (left columns red, right columns blue, both edges present) and the
    # whole height (the green band, 110 of 218 rows in file 10, covers the top ~32 of 64 rows, not more and not less)
    r, g, b = x[0, 0], x[0, 1], x[0, 2]
    assert float(r[40:, :28].mean()) > 0.9 and float(b[40:, :28].mean()) < -0.9
    assert float(b[40:, 36:].mean()) > 0.9 and float(r[40:, 36:].mean()) < -0.9
    assert float(g[:30].mean()) > 0.4 and float(g[35:].mean()) < -0.9
    # a non-square target keeps the [height, width] order of the config (celeba.yaml:17-19)
    tf2 = {"convert": True, "normalize": True, "resize": {"height": 64, "width": 48}}
    dm2 = CelebADataModule(str(tmp_path), 48, 64, 3, batch_size=2, num_workers=0, transforms=tf2)
    dm2.setup()
    x2, _ = next(iter(dm2.train_dataloader()))
    assert x2.shape[1:] == (3, 64, 48)
    with pytest.raises(FileNotFoundError):
        CelebADataModule(str(tmp_path / "nowhere"), 64, 64, 3, batch_size=4, num_workers=0, transforms=tf).prepare_data()


def test_resize_methods_follow_the_config(tmp_path):
    from PIL import Image
    from src.datamodules.base import ArrayImageDataset
    rng = np.random.default_rng(3)
    imgs = rng.integers(0, 256, (2, 28, 28, 1), dtype=np.uint8)
    for name, pil in (("nearest", Image.NEAREST), ("bilinear", Image.BILINEAR), ("bicubic", Image.BICUBIC), (None, Image.BICUBIC)):
        rz = {"height": 32, "width": 32}
        if name:
            rz["method"] = name
        ds = ArrayImageDataset(imgs, np.zeros(2, dtype=np.int64), {"convert": True, "normalize": False, "resize": rz})
        x, _ = ds[1]
        want = torch.from_numpy(np.asarray(Image.fromarray(imgs[1, :, :, 0]).resize((32, 32), pil))).float() / 255
        assert x.shape == (1, 32, 32) and torch.equal(x[0], want), name


def test_materialize_uint8_equals_per_access_chain(tmp_path):
    """The device-resident loader works on uint8 arrays with decode + resize applied once: materialising must give exactly the
    bytes the per-access chain sees, for the jpg folder (also through the worker pool) and for a resized array dataset."""
    from src.datamodules.base import ArrayImageDataset, materialize_uint8
    from src.datamodules.celeba import CelebADataModule
    _celeba_folder(tmp_path, n=9)
    tf = {"convert": True, "normalize": True, "resize": {"height": 32, "width": 32}}
    dm = CelebADataModule(str(tmp_path), 32, 32, 3, batch_size=4, num_workers=0, transforms=tf)
    dm.setup()
    for workers in (0, 2):
        mat = materialize_uint8(dm.train_data, workers)
        assert isinstance(mat, ArrayImageDataset) and mat.images.shape == (8, 32, 32, 3) and mat.images.dtype == np.uint8
        assert mat.resize is None and mat.normalize and not mat.flip
        for i in (0, 3, 7):
            assert torch.equal(mat[i][0], dm.train_data[i][0])
    rng = np.random.default_rng(0)
    ds = ArrayImageDataset(rng.integers(0, 256, (5, 28, 28, 1), dtype=np.uint8), np.arange(5), {"normalize": True, "resize": {"height": 32, "width": 32}})
    mat = materialize_uint8(ds)
    assert mat.images.shape == (5, 32, 32, 1) and all(torch.equal(mat[i][0], ds[i][0]) and mat[i][1] == ds[i][1] for i in range(5))
    assert materialize_uint8(mat) is mat
    # a resized ARRAY dataset through the worker pool (cifar10 / mnist with a transforms.resize override and num_workers > 0): the
    # per-item callable has to be picklable
    ds9 = ArrayImageDataset(rng.integers(0, 256, (9, 28, 28, 1), dtype=np.uint8), np.arange(9), {"normalize": True, "resize": {"height": 32, "width": 32}})
    mat9 = materialize_uint8(ds9, num_workers=2)
    assert mat9.images.shape == (9, 32, 32, 1) and all(torch.equal(mat9[i][0], ds9[i][0]) for i in range(9))
    # anything else is refused with a message, not an AttributeError
    import pytest
    with pytest.raises(TypeError, match="device_resident=False"):
        materialize_uint8([1, 2, 3])
