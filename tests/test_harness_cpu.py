"""Host harness on CPU: config composition, data modules, grid writer, trainer loop, DDP reducer."""
import json
import os
import pickle
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "image-generation-models_amd")


def test_compose_ddpm_cifar10():
    from src.runtime.config import Composer
    c = Composer(os.path.join(PKG, "configs")).compose("config", ["experiment=ddpm/cifar10", "model.hidden_dim=128",
                                                                  "+trainer.precision=bf16-mixed", "seed=7"])
    assert c.model._target_ == "src.models.ddpm.DDPM" and c.model.hidden_dim == 128 and c.model.dim_mults == [1, 2, 4]
    assert c.model.lr == 0.0001 and c.model.b1 == 0.9 and c.model.loss_type == "l1"        # configs/model/ddpm.yaml
    assert c.datamodule.batch_size == 128 and c.datamodule.transforms.normalize is True
    assert c.trainer.max_epochs == 100 and c.trainer.check_val_every_n_epoch == 10 and c.trainer.precision == "bf16-mixed"
    assert set(c.callbacks.keys()) == {"sample", "tqdm"}                                   # model swaps callbacks to ar_models
    assert c.exp_name == "ddpm/cifar10" and c.seed == 7
    assert c.data_dir.endswith("/data/") and "${" not in json.dumps(c)
    c3 = Composer(os.path.join(PKG, "configs")).compose("config", ["experiment=ddpm/celeba"])
    assert c3.model.dim_mults == [1, 2, 4, 8] and c3.datamodule.transforms.resize.width == 64


def test_instantiate_aliases_and_model():
    from src.runtime.config import Composer, instantiate
    c = Composer(os.path.join(PKG, "configs")).compose("config", ["experiment=ddpm/synthetic", "model.hidden_dim=8",
                                                                  "model.dim_mults=[1,2]"])
    model = instantiate(c.model, datamodule=c.datamodule, _recursive_=False)
    assert type(model).__name__ == "DDPM" and model.hparams.hidden_dim == 8
    trainer = instantiate(c.trainer, callbacks=[], logger=None)
    assert type(trainer).__name__ == "Trainer" and trainer.max_epochs == 1
    cbs = [instantiate(v) for v in c.callbacks.values()]
    assert {type(x).__name__ for x in cbs} == {"SampleImagesCallback", "ProgressBar"}


def test_cifar10_reader_and_normalisation(tmp_path):
    from src.datamodules.cifar10 import CIFAR10DataModule
    d = tmp_path / "cifar-10-batches-py"
    d.mkdir()
    rng = np.random.default_rng(0)
    for name, n in [(f"data_batch_{i}", 20) for i in range(1, 6)] + [("test_batch", 10)]:
        with open(d / name, "wb") as f:
            pickle.dump({"data": rng.integers(0, 256, (n, 3072), dtype=np.uint8), "labels": list(rng.integers(0, 10, n))}, f)
    dm = CIFAR10DataModule(str(tmp_path), 32, 32, 3, batch_size=16, num_workers=0, transforms={"convert": True, "normalize": True})
    dm.prepare_data(); dm.setup()
    assert len(dm.train_data) == 100 and len(dm.val_data) == 10
    x, y = next(iter(dm.val_dataloader()))
    assert x.shape == (10, 3, 32, 32) and x.dtype == torch.float32 and float(x.min()) >= -1 and float(x.max()) <= 1
    raw = pickle.load(open(d / "test_batch", "rb"))["data"][0].reshape(3, 32, 32)
    assert torch.allclose(x[0], (torch.from_numpy(raw).float() / 255 - 0.5) / 0.5)
    # data-parallel shards: disjoint, equal length, cover the set
    dm.set_shard(0, 2); a = [int(i) for i in dm._loader(dm.train_data, True).sampler]
    dm.set_shard(1, 2); b = [int(i) for i in dm._loader(dm.train_data, True).sampler]
    assert len(a) == len(b) == 50 and set(a) | set(b) == set(range(100)) and not set(a) & set(b)


def test_grid_matches_make_grid_layout():
    from src.callbacks.visualization import make_grid
    imgs = torch.linspace(-1, 1, 10 * 3 * 4 * 4).reshape(10, 3, 4, 4)
    g = make_grid(imgs, nrow=8, normalize=True, value_range=(-1, 1), pad_value=1)
    assert g.shape == (3, 2 * 6 + 2, 8 * 6 + 2)                 # torchvision: H*ymaps+padding, W*xmaps+padding
    assert float(g[:, 0, 0]) if False else torch.all(g[:, 0, :] == 1)
    assert torch.allclose(g[:, 2:6, 2:6], (imgs[0] + 1) / 2)
    assert torch.allclose(g[:, 8:12, 8:12], (imgs[9] + 1) / 2)   # second row, second column = image 9
    assert torch.all(g[:, 8:12, 14:18] == 1)                     # empty slot stays pad_value


class _Toy(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.lin = torch.nn.Linear(4, 1)
        self.trainer = None
        self.input_normalize = True
        self.seen_val = 0

    def training_step(self, batch, i):
        x, y = batch
        loss = ((self.lin(x).squeeze(-1) - y) ** 2).mean()
        self.trainer._log_metric("train_loss/loss", loss.detach())
        return loss

    def validation_step(self, batch, i):
        from src.models.base import ValidationResult
        self.seen_val += 1
        return ValidationResult(real_image=torch.zeros(4, 3, 4, 4), fake_image=torch.zeros(4, 3, 4, 4) if i == 0 else None)

    def configure_optimizers(self):
        return torch.optim.SGD(self.parameters(), lr=0.1)


def test_trainer_loop_callbacks_checkpoint(tmp_path, monkeypatch):
    from src.callbacks.visualization import SampleImagesCallback
    from src.runtime.loggers import TensorBoardLogger
    from src.runtime.trainer import Trainer
    monkeypatch.chdir(tmp_path)
    torch.manual_seed(0)
    x = torch.randn(64, 4); y = x @ torch.tensor([1.0, -2.0, 0.5, 3.0])
    ds = torch.utils.data.TensorDataset(x, y)
    loader = torch.utils.data.DataLoader(ds, batch_size=16)
    model = _Toy()
    tr = Trainer(accelerator="cpu", max_epochs=4, check_val_every_n_epoch=2, callbacks=[SampleImagesCallback()],
                 logger=TensorBoardLogger(str(tmp_path / "tb")), log_every_n_steps=2)
    tr.fit(model, train_dataloaders=loader, val_dataloaders=loader)
    assert tr.global_step == 16 and model.seen_val == 2 + 2 * 4          # 2 sanity batches + 2 validation epochs
    assert tr.callback_metrics["train_loss/loss"] < 0.5
    assert os.path.exists(tmp_path / "results" / "1.jpg") and os.path.exists(tmp_path / "results" / "3.jpg")
    ck = torch.load(tr.checkpoint_callback.best_model_path)
    assert set(ck["state_dict"]) == {"lin.weight", "lin.bias"} and ck["global_step"] == 16
    lines = open(tmp_path / "tb" / "metrics.jsonl").read().strip().splitlines()
    assert len(lines) >= 8 and "train_loss/loss" in json.loads(lines[0])


def _ddp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path[:0] = [ROOT, PKG]
    import torch.distributed as dist
    from src.runtime.ddp import FlatGradReducer, broadcast_parameters
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 100_000
    params = torch.full((n,), float(rank + 1))
    broadcast_parameters(params)
    flat = torch.arange(n, dtype=torch.float32) * (rank + 1)
    red = FlatGradReducer(flat, bucket_bytes=64 * 1024)
    red.begin()
    hi = n
    while hi > 0:                                  # backward finalises the buffer from the back, in uneven pieces
        lo = max(0, hi - 7919)
        red.range_ready(lo, hi); hi = lo
    red.finish()
    q.put((rank, float(params.sum()), flat.numpy().copy(), list(red.launched), red.grad_scale))   # by value: the worker may exit first
    dist.destroy_process_group()


def test_flat_grad_reducer_gloo_world2():
    """N=2 data-parallel path on CPU/gloo: bucketed all-reduce over the flat gradient buffer."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(30)
    n = 100_000
    want = torch.arange(n, dtype=torch.float32) * 3            # sum over ranks; Adam applies grad_scale = 1/2
    for rank, psum, flat, launched, scale in res:
        assert psum == n * 1.0                                   # parameters broadcast from rank 0
        assert torch.equal(torch.from_numpy(flat), want) and scale == 0.5
        assert launched[0][1] == n and launched[-1][0] == 0      # covers [0, n) back to front
        assert all(a[0] == b[1] for a, b in zip(launched, launched[1:]))
        assert len(launched) > 3                                 # really bucketed
        assert launched[-1][1] - launched[-1][0] <= 7919 + 4096  # the final, non-overlappable flush is small (tail rule)
        assert max(b - a for a, b in launched) >= 16384            # ... while the others are full buckets


def _shared_materialize_worker(rank, world, port, tmp):
    import numpy as np
    import torch.distributed as dist
    from src.datamodules import base as B
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(5)
        imgs = rng.integers(0, 256, (12, 10, 14, 3), dtype=np.uint8)
        ds = B.ArrayImageDataset(imgs, np.arange(12), {"normalize": True, "flip": None, "resize": {"width": 8, "height": 6}})
        calls = []
        orig = ds.resized
        ds.resized = lambda im: (calls.append(1), orig(im))[1]
        out = B.materialize_uint8_shared(ds, 0, local_rank=rank)
        assert out.images.shape == (12, 6, 8, 3) and out.images.dtype == np.uint8 and out.normalize and not out.flip
        assert len(calls) == (12 if rank == 0 else 0)            # decoded / resized ONCE, by local rank 0
        np.save(os.path.join(tmp, f"r{rank}.npy"), out.images)
    finally:
        dist.destroy_process_group()


def test_device_resident_dataset_is_materialized_once_per_node_gloo_world2(tmp_path):
    """Data parallelism, device-resident dataset: local rank 0 decodes and resizes, the other rank reads the array from /dev/shm
    (src/datamodules/base.py::materialize_uint8_shared); both end up with the same bytes and the hand-over files are gone."""
    import numpy as np
    import torch.multiprocessing as mp
    mp.spawn(_shared_materialize_worker, args=(2, 32500 + os.getpid() % 2000, str(tmp_path)), nprocs=2, join=True)
    a, b = np.load(tmp_path / "r0.npy"), np.load(tmp_path / "r1.npy")
    assert np.array_equal(a, b) and a.shape == (12, 6, 8, 3)
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("mi_ddpm_u8_")]


def test_compose_vqvae_and_multi_buffer_reducer_gloo_world2():
    """experiment=vqvae/cifar10 composes to the reference's values; the trainer's reducer group averages several flat
    gradient buffers across 2 gloo ranks."""
    from src.runtime.config import Composer
    c = Composer(os.path.join(PKG, "configs")).compose("config", ["experiment=vqvae/cifar10"])
    assert c.model._target_ == "src.models.vqvae.VQVAE" and c.model.latent_dim == 64 and c.model.beta == 0.25 and c.model.lr == 0.001
    assert c.model.encoder._target_ == "src.networks.vqvae.Encoder" and c.model.decoder._target_ == "src.networks.vqvae.Decoder"
    assert c.model.encoder.input_channel is None and c.datamodule.batch_size == 128 and c.exp_name == "vqvae/cifar10"
    import torch.multiprocessing as mp
    mp.spawn(_reducer_group_worker, args=(2, 31500 + os.getpid() % 2000), nprocs=2, join=True)


def _reducer_group_worker(rank, world, port):
    import torch.distributed as dist
    from src.runtime.ddp import FlatGradReducer
    from src.runtime.trainer import _ReducerGroup
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        bufs = [torch.full((n,), float(rank + 1)) * torch.arange(1, n + 1) for n in (1000, 37, 4096)]
        grp = _ReducerGroup([FlatGradReducer(b) for b in bufs])
        grp.begin()
        grp.finish()
        for b in bufs:
            assert torch.equal(b, 3.0 * torch.arange(1, b.numel() + 1))          # SUM over ranks (1 + 2); the optimizer divides
        assert grp.grad_scale == 0.5
    finally:
        dist.destroy_process_group()


class _ToyManual(__import__("src.runtime.lightning_lite", fromlist=["LightningModule"]).LightningModule):
    """Two optimizers stepped by the model itself (the GAN pattern of src/models/wgan_gp.py:52-107)."""

    def __init__(self):
        super().__init__()
        self.automatic_optimization = False
        self.a = torch.nn.Linear(4, 1)
        self.b = torch.nn.Linear(4, 1)
        self.steps = [0, 0]

    def configure_optimizers(self):
        return torch.optim.SGD(self.a.parameters(), lr=0.1), torch.optim.SGD(self.b.parameters(), lr=0.1)

    def training_step(self, batch, i):
        x, y = batch
        opt_a, opt_b = self.optimizers()
        which = i % 3 == 2
        net, opt = (self.b, opt_b) if which else (self.a, opt_a)
        loss = ((net(x).squeeze(-1) - y) ** 2).mean()
        opt.zero_grad()
        self.manual_backward(loss)
        opt.step()
        self.steps[which] += 1
        self.log("train_loss/which", float(which))


class _ToySched(_Toy):
    def configure_optimizers(self):
        opt = torch.optim.SGD(self.parameters(), lr=0.1)
        return [opt], [torch.optim.lr_scheduler.StepLR(opt, 1, gamma=0.5)]


def test_trainer_manual_optimization_and_schedulers(tmp_path, monkeypatch):
    """The two Lightning conventions the widened models rely on: `automatic_optimization = False` with `self.optimizers()` /
    `manual_backward` (WGAN-GP) and `configure_optimizers` returning ([optimizers], [schedulers]) (VAE: StepLR per epoch)."""
    from src.runtime.trainer import Trainer
    monkeypatch.chdir(tmp_path)
    torch.manual_seed(0)
    x = torch.randn(48, 4); y = x @ torch.tensor([1.0, -2.0, 0.5, 3.0])
    loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(x, y), batch_size=8)
    m = _ToyManual()
    a0, b0 = m.a.weight.detach().clone(), m.b.weight.detach().clone()
    tr = Trainer(accelerator="cpu", max_epochs=2, enable_checkpointing=True, num_sanity_val_steps=0)
    tr.fit(m, train_dataloaders=loader)
    assert tr.global_step == 12 and m.steps == [8, 4]                       # the trainer never stepped anything itself
    assert not torch.equal(m.a.weight, a0) and not torch.equal(m.b.weight, b0)
    ck = torch.load(tr.checkpoint_callback.best_model_path)
    assert len(ck["optimizer_states"]) == 2
    s = _ToySched()
    tr = Trainer(accelerator="cpu", max_epochs=3, enable_checkpointing=False, num_sanity_val_steps=0)
    tr.fit(s, train_dataloaders=loader)
    assert abs(tr.optimizer.param_groups[0]["lr"] - 0.1 * 0.5 ** 3) < 1e-12   # one scheduler step per epoch


def test_bench_self_launches_ranks_without_torchrun():
    """`python bench.py --gpus 2` with no launcher in the environment starts two ranks itself (torch.distributed.run, 127.0.0.1)
    and rank 0 prints ONE line with n_gpus = 2 and the number of ranks the backend really connected.  The host dry run swaps
    RCCL for gloo and the kernels for a stand-in step; launch, rendezvous, barrier and max-over-ranks code is the real one."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run-cpu", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["self_launched"] and out["dry_run"] and out["steps"] == 3
    assert out["buckets_bytes"] and out["scaling"] == "weak"
    # round 5 (what the first 8-GPU minute must answer by itself): per-rank step times, the step three ways
    sp = out["rank_ms_per_step"]
    assert len(sp["per_rank"]) == 2 and sp["min"] <= sp["max"] and abs(sp["max"] - out["ms_per_step"]) < 1.0
    ways = out["comm"]["ways_ms"]
    assert set(ways) == {"segmented_graph", "eager_overlap", "eager_allreduce_after_backward", "exposed_comm_ms"}
    assert all(isinstance(v, float) for v in ways.values())


def test_bench_dry_run_n1_runs_the_data_parallel_path_and_the_channel_sweep_relaunches():
    """N = 1 also reports a leg through the data-parallel code path (one-rank group); `--sweep-channels` runs the N-rank bench once
    per NCCL_MAX_NCHANNELS in {default, 8, 16} and prints ONE line that carries all three."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "NCCL_MAX_NCHANNELS")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--dry-run-cpu", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["dp_path_n1"]["rccl_ranks"] == 1 and out["dp_path_n1"]["ms_per_step"] > 0
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run-cpu", "--sweep-channels", "--steps", "2",
                        "--warmup", "1"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    sw = out["channel_sweep"]
    assert [e["NCCL_MAX_NCHANNELS"] for e in sw] == ["default", "8", "16"] and all("error" not in e and e["ms_per_step"] > 0 for e in sw), sw
    assert out["n_gpus"] == 2 and out["comm"]["NCCL_MAX_NCHANNELS"] is None


def test_wgrad_queue_flushed_is_a_prefix_not_a_count():
    """WgradQueue.flushed releases gradient ranges to the all-reduce: it must be the longest issued PREFIX of the pushed layers
    (the 3x3 and 1x1 kinds flush independently), not the number of issued layers."""
    from src.ops.functional import WgradQueue
    q = WgradQueue(group=8)
    q.pushed, q._seq3, q._seq1 = 5, [1, 4], [2, 3, 5]
    assert q.flushed == 0
    q._seq1 = []                         # every 1x1 layer is out, layer 1 (3x3) is not
    assert q.flushed == 0
    q._seq3, q._seq1 = [4], [5]
    assert q.flushed == 3
    q._seq3 = []
    assert q.flushed == 4
    q._seq1 = []
    assert q.flushed == 5
