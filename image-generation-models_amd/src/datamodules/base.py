"""Data modules without torchvision: raw dataset files -> uint8 arrays -> `(x/255 - 0.5)/0.5`
float tensors in NCHW (reference pipeline: ToTensor + Normalize(0.5, 0.5),
src/datamodules/base.py:37-71).  Two ways to serve them:

  * `DeviceBatchLoader` (default on a GPU): the whole uint8 dataset sits in HBM (CIFAR-10 150 MB, CelebA at 64x64 2.5 GB, of
    288 GB) and one HIP launch per batch gathers the rows and applies ToTensor / flip / Normalize (`mi_u8_gather_normalize`):
    no host work, no PCIe traffic per step;
  * torch DataLoaders with the reference's settings (shuffle on train, fork workers, no drop_last, base.py:14-27), plus pinned
    host batches and a side-stream prefetch to the device (`DevicePrefetcher`) -- for datasets that do not fit or
    `datamodule.device_resident=false`.

Under data-parallel training each rank reads a disjoint, equally sized shard per epoch (DistributedSampler semantics)."""
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

    def resized(self, img: np.ndarray) -> np.ndarray:
        """transforms.Resize([height, width], method) on a PIL image (reference base.py:43-49): PIL's resize takes (width, height);
        the default method is BICUBIC; no crop."""
        if self.resize is None:
            return img
        from PIL import Image
        size = (int(_cfg_get(self.resize, "width")), int(_cfg_get(self.resize, "height")))
        method = {"nearest": Image.NEAREST, "bilinear": Image.BILINEAR, "bicubic": Image.BICUBIC}.get(_cfg_get(self.resize, "method"), Image.BICUBIC)
        img = np.asarray(Image.fromarray(img.squeeze()).resize(size, method))
        return img[:, :, None] if img.ndim == 2 else img

    def __getitem__(self, i):
        img = self.resized(self.images[i])
        x = torch.from_numpy(np.ascontiguousarray(img)).permute(2, 0, 1).float().div_(255.0)     # ToTensor
        if self.flip and torch.rand(()) < 0.5:
            x = x.flip(-1)
        if self.normalize:
            x = (x - 0.5) / 0.5
        return x, int(self.labels[i])


def _resized_item(dataset, i):
    """Module-level so that multiprocessing can pickle `functools.partial(_resized_item, dataset)` (a local lambda cannot be)."""
    return dataset.resized(dataset.images[i])


def materialize_uint8(dataset, num_workers: int = 0) -> "ArrayImageDataset":
    """Apply the deterministic part of the transform chain (decode, resize) ONCE and return an ArrayImageDataset over a uint8
    array without a resize step; flip / normalize stay per-access.  Accepts ArrayImageDataset (with or without resize) and
    datasets exposing `load_uint8(i)` + `_tf` (the jpg folder); anything else raises TypeError."""
    if isinstance(dataset, ArrayImageDataset) and dataset.resize is None:
        return dataset
    n = len(dataset)
    if isinstance(dataset, ArrayImageDataset):
        import functools
        tf, labels = dataset, np.asarray(dataset.labels)
        get = functools.partial(_resized_item, dataset)
    elif hasattr(dataset, "load_uint8") and hasattr(dataset, "_tf"):
        tf, labels = dataset._tf, np.zeros(n, dtype=np.int64)
        get = dataset.load_uint8
    else:
        raise TypeError(f"materialize_uint8: {type(dataset).__name__} is neither an ArrayImageDataset nor a dataset with "
                        "load_uint8(i) / _tf; use device_resident=False (DataLoader + DevicePrefetcher) for it")
    if num_workers > 0 and n >= 4 * num_workers:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(num_workers) as pool:
            imgs = pool.map(get, range(n), chunksize=max(1, n // (num_workers * 16)))
    else:
        imgs = [get(i) for i in range(n)]
    arr = np.stack(imgs) if n else np.zeros((0, 1, 1, 1), dtype=np.uint8)
    out = ArrayImageDataset(arr, labels, None)
    out.normalize, out.flip = tf.normalize, tf.flip
    return out


def materialize_uint8_shared(dataset, num_workers: int = 0, local_rank: int = 0) -> "ArrayImageDataset":
    """materialize_uint8 under data parallelism: ONE rank per node (local rank 0) decodes / resizes the dataset and leaves the
    uint8 array in /dev/shm; the node's other ranks read it from there instead of repeating the work (CelebA: 202 599 jpgs decoded
    and resized once instead of once per GPU, on the same host cores).  Needs an initialised process group (the name of the file is
    agreed on by a broadcast, the hand-over is two barriers); without one it is materialize_uint8."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return materialize_uint8(dataset, num_workers)
    if isinstance(dataset, ArrayImageDataset) and dataset.resize is None:
        return dataset                                     # nothing to decode: every rank already holds the array it read
    import os
    import socket
    import uuid
    # node-local rank from the process group itself (LOCAL_RANK is absent under manual RANK / WORLD_SIZE launches, mp.spawn and some
    # Slurm set-ups; falling back to the GLOBAL rank left nodes >= 1 without a writer): the first rank on each host writes
    hosts = [None] * dist.get_world_size()
    dist.all_gather_object(hosts, socket.gethostname())
    rank = dist.get_rank()
    writer = hosts.index(hosts[rank]) == rank
    tok = [uuid.uuid4().hex if rank == 0 else None]
    dist.broadcast_object_list(tok, src=0)
    base = os.path.join("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp", f"mi_ddpm_u8_{tok[0]}")
    out, err = None, None
    if writer:
        try:
            out = materialize_uint8(dataset, num_workers)
            np.save(base + "_x.npy", out.images)
            np.save(base + "_y.npy", np.asarray(out.labels))
        except Exception as exc:      # noqa: BLE001  (reported to every rank below: a bare barrier would leave the peers waiting forever)
            err = f"{type(exc).__name__}: {exc}"
    errs = [None] * dist.get_world_size()
    dist.all_gather_object(errs, err)                       # the files are complete -- or every rank learns why not
    try:
        bad = [e for e in errs if e]
        if bad:
            raise RuntimeError(f"materialize_uint8_shared: the decoding rank failed ({bad[0]})")
        if out is None:
            tf = dataset if isinstance(dataset, ArrayImageDataset) else dataset._tf
            if os.path.exists(base + "_x.npy") and os.path.exists(base + "_y.npy"):
                out = ArrayImageDataset(np.load(base + "_x.npy"), np.load(base + "_y.npy"), None)
                out.normalize, out.flip = tf.normalize, tf.flip
            else:                                           # (hosts that share a name but not /dev/shm: decode here after all)
                out = materialize_uint8(dataset, num_workers)
    finally:
        dist.barrier()                                      # every rank has read them
        if writer:
            for suffix in ("_x.npy", "_y.npy"):
                try:
                    os.remove(base + suffix)
                except OSError:
                    pass
    return out


class DeviceBatchLoader:
    """Serves (images fp32 NCHW, labels int64) batches from a uint8 dataset resident in HBM.  Epoch order: a fresh random
    permutation per epoch seeded from torch's global CPU generator with the draw torch's RandomSampler makes
    (`int(torch.empty((), dtype=torch.int64).random_())`), or the ShardSampler order under data-parallel training; the last
    batch is ragged (no drop_last), as with the reference's DataLoader.  NOT the reference's sample order for the same global seed:
    a torch DataLoader iterator also draws its `_base_seed` from the global generator before the sampler's seed each epoch, and the
    flips here come from the device generator.  Under world > 1 every rank holds the whole uint8 dataset in its HBM (it draws its shard
    from it); the decode / resize behind it runs once per node (materialize_uint8_shared)."""

    def __init__(self, dataset: ArrayImageDataset, batch_size: int, device, shuffle: bool, sampler=None):
        if dataset.resize is not None:
            raise ValueError("materialize_uint8() the dataset first: resize cannot run per access on the device")
        self.device = torch.device(device)
        self.data = torch.from_numpy(np.ascontiguousarray(dataset.images)).to(self.device)        # uint8 [N,H,W,C]
        self.labels = torch.from_numpy(np.asarray(dataset.labels, dtype=np.int64)).to(self.device)
        self.normalize, self.flip = dataset.normalize, dataset.flip
        self.batch_size, self.shuffle, self.sampler = int(batch_size), shuffle, sampler
        self.n = len(dataset)

    def __len__(self):
        n = len(self.sampler) if self.sampler is not None else self.n
        return (n + self.batch_size - 1) // self.batch_size

    def _order(self) -> torch.Tensor:
        if self.sampler is not None:
            return torch.tensor(list(iter(self.sampler)), dtype=torch.int64)
        if self.shuffle:
            seed = int(torch.empty((), dtype=torch.int64).random_().item())
            return torch.randperm(self.n, generator=torch.Generator().manual_seed(seed))
        return torch.arange(self.n)

    def __iter__(self):
        from src.ops import functional as K
        order = self._order().to(self.device)
        for lo in range(0, order.numel(), self.batch_size):
            idx = order[lo:lo + self.batch_size]
            flip = (torch.rand(idx.numel(), device=self.device) < 0.5) if self.flip else None
            yield K.u8_gather_normalize(self.data, idx, flip, self.normalize), self.labels[idx]


class DevicePrefetcher:
    """Wraps a (pinned-memory) DataLoader: the next batch's host->device copy runs on a side stream while the current step
    computes; the consumer's stream waits on the copy's event only."""

    def __init__(self, loader, device):
        self.loader, self.device = loader, torch.device(device)
        self.stream = torch.cuda.Stream(self.device)

    def __len__(self):
        return len(self.loader)

    def _put(self, batch):
        with torch.cuda.stream(self.stream):
            return tuple(b.to(self.device, non_blocking=True) if torch.is_tensor(b) else b for b in batch)

    def __iter__(self):
        it = iter(self.loader)
        try:
            nxt = self._put(next(it))
        except StopIteration:
            return
        while nxt is not None:
            torch.cuda.current_stream(self.device).wait_stream(self.stream)
            cur = nxt
            for b in cur:
                if torch.is_tensor(b):
                    b.record_stream(torch.cuda.current_stream(self.device))
            try:
                nxt = self._put(next(it))
            except StopIteration:
                nxt = None
            yield cur


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
    # uint8 datasets up to this size are kept in HBM when device_resident is "auto"
    DEVICE_RESIDENT_MAX_BYTES = 32 << 30

    def __init__(self, width, height, channels, batch_size, num_workers, device_resident="auto"):
        super().__init__()
        self.width, self.height, self.channels = width, height, channels
        self.batch_size, self.num_workers = batch_size, num_workers
        self.device_resident = device_resident          # "auto" | True | False  (datamodule.device_resident=...)
        self._rank, self._world = 0, 1
        self._train_sampler = None
        self._device = None
        self._resident = {}

    def bind_device(self, device):
        """Called by the Trainer once the rank's device is known: from here on the loaders deliver device batches."""
        self._device = torch.device(device) if device is not None and torch.device(device).type == "cuda" else None

    def set_shard(self, rank, world):
        self._rank, self._world = rank, world

    def set_epoch(self, epoch):
        if self._train_sampler is not None:
            self._train_sampler.set_epoch(epoch)

    def _want_resident(self, data) -> bool:
        if self._device is None or self.device_resident in (False, "false", "False", 0):
            return False
        if self.device_resident in (True, "true", "True", 1):
            return True
        if isinstance(data, ArrayImageDataset):
            n, per = len(data), self.width * self.height * self.channels
        elif hasattr(data, "load_uint8"):
            n, per = len(data), self.width * self.height * self.channels
        else:
            return False
        return n * per <= self.DEVICE_RESIDENT_MAX_BYTES

    def _loader(self, data, shuffle):
        sampler = None
        if self._world > 1:
            sampler = ShardSampler(len(data), self._rank, self._world, shuffle=shuffle)
            if shuffle:
                self._train_sampler = sampler
        if self._want_resident(data):
            key = id(data)
            if key not in self._resident:
                import os
                self._resident[key] = (materialize_uint8_shared(data, self.num_workers, int(os.environ.get("LOCAL_RANK", self._rank)))
                                       if self._world > 1 else materialize_uint8(data, self.num_workers))
            return DeviceBatchLoader(self._resident[key], self.batch_size, self._device, shuffle, sampler)
        kw = dict(batch_size=self.batch_size, num_workers=self.num_workers)
        if self.num_workers > 0:
            kw["multiprocessing_context"] = "fork"
            kw["persistent_workers"] = True
        if self._device is not None:
            kw["pin_memory"] = True
        loader = DataLoader(data, sampler=sampler, **kw) if sampler is not None else DataLoader(data, shuffle=shuffle, **kw)
        return DevicePrefetcher(loader, self._device) if self._device is not None else loader

    def train_dataloader(self):
        return self._loader(self.train_data, True)

    def val_dataloader(self):
        return self._loader(self.val_data, False)
