"""A Lightning-subset `Trainer` for the DDPM path (used when pytorch_lightning is absent).

Reproduces what the reference relies on from `pytorch_lightning.Trainer`:
automatic optimisation (zero_grad / training_step / backward / step), device placement, two
sanity-validation batches, `check_val_every_n_epoch`, `on_validation_batch_end` callbacks
receiving `validation_step`'s return value, scalar logging every `log_every_n_steps`,
checkpoints with the reference's state_dict keys, and data-parallel training with one process
per GPU (`devices=N` under torchrun): here the gradient all-reduce runs over the UNet's flat
gradient buffer through RCCL, bucketed and overlapped with backward (src/runtime/ddp.py).
`precision="bf16-mixed"` selects the bf16-MFMA kernels (fp32 master weights, fp32 accumulate).
"""
from __future__ import annotations

import os
import time
from typing import Any, Dict, List, Optional

import torch
import torch.distributed as dist

from .lightning_lite import Callback


class ProgressBar(Callback):
    """Stand-in for TQDMProgressBar(refresh_rate=N): prints a line every N steps on rank 0."""

    def __init__(self, refresh_rate: int = 5, **_):
        self.refresh_rate = max(1, int(refresh_rate))


class _Checkpoint:
    def __init__(self):
        self.best_model_path = ""


class _ReducerGroup:
    """begin / finish / grad_scale over several FlatGradReducers (no overlap hooks: finish() reduces each buffer whole)."""

    def __init__(self, reducers):
        self.reducers = reducers

    @property
    def grad_scale(self):
        return self.reducers[0].grad_scale

    def begin(self):
        for r in self.reducers:
            r.begin()

    def finish(self):
        for r in self.reducers:
            r.range_ready(0, r.flat.numel())           # the whole buffer is final once backward returned
            r.finish()


class Trainer:
    def __init__(self, devices=1, max_epochs: int = 20, check_val_every_n_epoch: int = 1, enable_model_summary: bool = False,
                 callbacks: Optional[List[Callback]] = None, logger=None, precision: str = "32", accelerator: str = "auto",
                 max_steps: int = -1, limit_train_batches: Optional[int] = None, limit_val_batches: Optional[int] = None,
                 num_sanity_val_steps: int = 2, log_every_n_steps: int = 50, enable_checkpointing: bool = True,
                 default_root_dir: str = ".", fast_dev_run: bool = False, strategy: str = "auto", graph_step: Optional[bool] = None, **unused):
        # graph_step (not a Lightning argument; `+trainer.graph_step=true|false`): replay training_step + backward + optimizer step as one
        # hipGraph (src/runtime/graphed.py); under data parallelism (DDPM) as a chain of graphs cut at the gradient buckets with the
        # all-reduces issued between them.  Automatic optimization, fused Adam only.  Default (None) = wherever the step can be
        # captured, with the eager loop as the fallback if the capture fails: the replay is never slower than eager by more than noise
        # (cfg 2 at B = 128: 23.9 vs 24.0 k images/s) and wins wherever the host is the bottleneck (cfg 3 at its per-GPU batch of 32:
        # 6.57 vs 5.30 k images/s, gpurun_out/graph_ab.txt of round 4); true = insist (a failing capture raises), false = eager.
        self.graph_step = None if graph_step is None else bool(graph_step)
        self.devices = devices
        self.max_epochs, self.max_steps = max_epochs, max_steps
        self.check_val_every_n_epoch = check_val_every_n_epoch
        self.callbacks = list(callbacks or [])
        self.logger = logger
        self.precision = str(precision)
        self.accelerator = accelerator
        self.limit_train_batches, self.limit_val_batches = limit_train_batches, limit_val_batches
        self.num_sanity_val_steps = num_sanity_val_steps
        self.log_every_n_steps = log_every_n_steps
        self.enable_checkpointing = enable_checkpointing
        self.default_root_dir = default_root_dir
        if fast_dev_run:
            self.max_epochs, self.limit_train_batches, self.limit_val_batches, self.num_sanity_val_steps = 1, 1, 1, 0
        self.current_epoch = 0
        self.global_step = 0
        self.sanity_checking = False
        self.callback_metrics: Dict[str, Any] = {}
        self.checkpoint_callback = _Checkpoint()
        self.world_size = int(os.environ.get("WORLD_SIZE", "1"))
        self.global_rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self._pending: Dict[str, Any] = {}
        self._reducer = None

    # ------------------------------------------------------------------ plumbing
    @property
    def is_global_zero(self) -> bool:
        return self.global_rank == 0

    def _device(self) -> torch.device:
        want_cpu = self.accelerator == "cpu" or self.devices in (0, "0")
        if want_cpu or not torch.cuda.is_available():
            return torch.device("cpu")
        # one process per GPU: rank r of the node drives device r (replaces the nvidia-smi picker of train.py:44-45).
        # MI_DDPM_ONE_GPU_RANKS=1 (with MI_DIST_BACKEND=gloo): every rank on device 0 -- the data-parallel path on a 1-GPU box
        if os.environ.get("MI_DDPM_ONE_GPU_RANKS") == "1":
            return torch.device("cuda", 0)
        return torch.device("cuda", self.local_rank)

    def _log_metric(self, name, value):
        self._pending[name] = value

    def _flush_metrics(self):
        if not self._pending:
            return
        vals = {k: (float(v) if torch.is_tensor(v) else v) for k, v in self._pending.items()}   # the only host sync
        self.callback_metrics.update(vals)
        if self.logger is not None and self.is_global_zero:
            self.logger.log_metrics(vals, step=self.global_step)
        self._pending.clear()

    def _to_device(self, batch, device):
        if torch.is_tensor(batch):
            return batch.to(device, non_blocking=True)
        if isinstance(batch, (list, tuple)):
            return type(batch)(self._to_device(b, device) for b in batch)
        return batch

    def _setup_distributed(self, model, device):
        if self.world_size <= 1:
            return
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            # nccl == RCCL on ROCm.  MI_DIST_BACKEND=gloo: several ranks on ONE GPU (RCCL refuses that) -- how the data-parallel
            # Trainer path is exercised on a single-GPU box (tests/test_ddp_gpu.py)
            backend = os.environ.get("MI_DIST_BACKEND") or ("nccl" if device.type == "cuda" else "gloo")
            dist.init_process_group(backend, device_id=device if (device.type == "cuda" and backend == "nccl") else None)
        from .ddp import FlatGradReducer, broadcast_parameters
        net = getattr(model, "denoising_model", None)
        if net is not None and hasattr(net, "flat_grads"):
            broadcast_parameters(net.flat_params)
            self._reducer = FlatGradReducer(net.flat_grads)
            net.grad_ready_hook = self._reducer.range_ready
        elif hasattr(model, "flat_nets"):              # several small flat buffers (VQ-VAE): reduced when backward is done
            for n in model.flat_nets():
                broadcast_parameters(n.flat_params)
            if getattr(model, "automatic_optimization", True):     # manual-optimization models reduce their own gradients
                self._reducer = _ReducerGroup([FlatGradReducer(n.flat_grads) for n in model.flat_nets()])

    # ------------------------------------------------------------------ loops
    def fit(self, model, datamodule=None, train_dataloaders=None, val_dataloaders=None):
        device = self._device()
        if device.type == "cuda":
            torch.cuda.set_device(device)      # kernels launch on the CURRENT HIP device / its current stream (one process per GPU)
        model.trainer = self
        model.to(device)
        if self.precision.startswith("bf16"):
            for m in model.modules():
                if hasattr(m, "compute_mode"):
                    m.compute_mode = "bf16"
        self._setup_distributed(model, device)
        if datamodule is not None:
            if self.is_global_zero:
                datamodule.prepare_data()
            if self.world_size > 1:
                dist.barrier()
            datamodule.setup("fit")
            if self.world_size > 1 and hasattr(datamodule, "set_shard"):
                datamodule.set_shard(self.global_rank, self.world_size)    # DistributedSampler semantics
            if hasattr(datamodule, "bind_device"):
                datamodule.bind_device(device)                             # device-resident dataset / pinned prefetch
            train_dataloaders = datamodule.train_dataloader()
            val_dataloaders = datamodule.val_dataloader()
        optimizer = model.configure_optimizers()
        schedulers = []
        if (isinstance(optimizer, (list, tuple)) and len(optimizer) == 2 and all(isinstance(o, (list, tuple)) for o in optimizer)):
            optimizer, schedulers = list(optimizer[0]), list(optimizer[1])     # Lightning's ([optimizers], [lr_schedulers]) form
            if len(optimizer) == 1:
                optimizer = optimizer[0]
        self.lr_schedulers = schedulers
        manual = not getattr(model, "automatic_optimization", True)
        if manual:                                     # the model steps its own optimizers (GANs); it also averages its gradients
            opts = list(optimizer) if isinstance(optimizer, (list, tuple)) else [optimizer]
            object.__setattr__(model, "_optimizers", opts)
            for o in opts:
                if self.world_size > 1 and hasattr(o, "grad_scale"):
                    o.grad_scale = 1.0 / self.world_size
        elif self._reducer is not None and hasattr(optimizer, "grad_scale"):
            optimizer.grad_scale = self._reducer.grad_scale
        self.optimizer = optimizer
        from .optim import FlatAdam
        from .ddp import FlatGradReducer
        # one graph per step without a reducer; under data parallelism the capture is cut at the gradient buckets (graphed.py), which
        # needs the autograd-free step of the DDPM module and the overlapping single-buffer reducer
        seg_graph = (isinstance(self._reducer, FlatGradReducer) and hasattr(model, "training_step_and_backward"))
        use_graph = (self.graph_step is not False and not manual and isinstance(optimizer, FlatAdam) and device.type == "cuda"
                     and (self._reducer is None or seg_graph))
        if self.graph_step and not use_graph:
            raise RuntimeError("trainer.graph_step=true needs automatic optimization, the fused FlatAdam, a GPU and (under data "
                               "parallelism) the DDPM module's autograd-free step with the flat-gradient reducer")
        gstep = None
        if use_graph:
            optimizer.device_state = True
        bar = next((c for c in self.callbacks if isinstance(c, ProgressBar)), None)

        if self.num_sanity_val_steps and val_dataloaders is not None:
            self.sanity_checking = True
            self._validate(model, val_dataloaders, device, limit=self.num_sanity_val_steps)
            self.sanity_checking = False

        done = False
        for epoch in range(self.max_epochs):
            self.current_epoch = epoch
            model.train()
            if datamodule is not None and hasattr(datamodule, "set_epoch"):
                datamodule.set_epoch(epoch)
            t0 = time.time()
            for i, batch in enumerate(train_dataloaders):
                if self.limit_train_batches is not None and i >= self.limit_train_batches:
                    break
                batch = self._to_device(batch, device)
                if manual:
                    model.training_step(batch, i)
                elif use_graph and gstep is not None and batch[0].shape == gstep.x.shape:
                    gstep(batch)
                    for k_, v_ in gstep.logged.items():
                        self._log_metric(k_, v_)
                elif use_graph and self.global_step >= 1 and gstep is None:
                    # step 0 ran eagerly (lazy module loads, workspaces, Adam state); capture on this batch and replay it once
                    from .graphed import GraphedTrainStep, SegmentedGraphedTrainStep
                    model._logged.clear()
                    cap_exc = None
                    try:
                        gstep = (SegmentedGraphedTrainStep(model, optimizer, self._reducer, batch) if self._reducer is not None
                                 else GraphedTrainStep(model, optimizer, batch, warmup=0))
                    except Exception as exc:      # noqa: BLE001  (automatic mode: an uncapturable step stays eager, loudly)
                        cap_exc = exc
                    # every rank must run the same KIND of step (the replayed segments and the eager autograd path issue their bucket
                    # all-reduces from different places): one failed capture anywhere puts every rank on the eager path
                    failed = self._any_rank(cap_exc is not None, device)
                    if failed:
                        if self.graph_step:
                            raise cap_exc if cap_exc is not None else RuntimeError("hipGraph capture failed on another rank")
                        why = f"{type(cap_exc).__name__}: {cap_exc}" if cap_exc is not None else "failed on another rank"
                        print(f"[trainer] hipGraph capture of the training step failed ({why}); continuing eagerly", flush=True)
                        use_graph, gstep = False, None
                        optimizer.device_state = False          # the eager path keeps its step count and lr on the host again
                        torch.cuda.synchronize()
                        # what the aborted capture recorded never ran: derived state keyed on the parameters (bf16 weight copies, ...) is
                        # stale although its keys say otherwise -- rebuild it before the eager step reads it
                        for n_ in getattr(optimizer, "nets", []):
                            n_.mark_params_dirty()
                        optimizer.zero_grad()
                        if self._reducer is not None:
                            self._reducer.begin()
                        loss = model.training_step(batch, i)
                        loss.backward()
                        if self._reducer is not None:
                            self._reducer.finish()
                        optimizer.step()
                    else:
                        gstep.logged = dict(model._logged)
                        gstep(batch)
                        for k_, v_ in gstep.logged.items():
                            self._log_metric(k_, v_)
                else:
                    optimizer.zero_grad()
                    if self._reducer is not None:
                        self._reducer.begin()
                    loss = model.training_step(batch, i)
                    loss.backward()
                    if self._reducer is not None:
                        self._reducer.finish()
                    optimizer.step()
                self.global_step += 1
                if self.global_step % self.log_every_n_steps == 0:
                    self._flush_metrics()
                if bar is not None and self.is_global_zero and self.global_step % bar.refresh_rate == 0:
                    rate = (i + 1) / max(time.time() - t0, 1e-9)
                    print(f"\repoch {epoch} step {i + 1} ({rate:.1f} it/s)", end="", flush=True)
                if 0 < self.max_steps <= self.global_step:
                    done = True
                    break
            self._flush_metrics()
            if bar is not None and self.is_global_zero:
                print()
            for sch in self.lr_schedulers:                     # epoch-interval schedulers (Lightning's default)
                sch.step()
            if use_graph:
                optimizer.sync_lr()                            # the captured Adam reads its learning rate from device memory
            for cb in self.callbacks:
                cb.on_train_epoch_end(self, model)
            if val_dataloaders is not None and (epoch + 1) % self.check_val_every_n_epoch == 0:
                self._validate(model, val_dataloaders, device, limit=self.limit_val_batches)
                if self.enable_checkpointing and self.is_global_zero:
                    self.save_checkpoint(os.path.join(self.default_root_dir, "checkpoints", f"epoch={epoch}-step={self.global_step}.ckpt"), model)
            if done:
                break
        if self.enable_checkpointing and self.is_global_zero and not self.checkpoint_callback.best_model_path:
            self.save_checkpoint(os.path.join(self.default_root_dir, "checkpoints", "last.ckpt"), model)

    @staticmethod
    def _any_rank(flag: bool, device) -> bool:
        """True when `flag` holds on ANY rank (a MAX all-reduce; single process: the flag itself)."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return bool(flag)
        t = torch.tensor([1 if flag else 0], device=device if dist.get_backend() == "nccl" else "cpu", dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return bool(int(t.item()))

    @torch.no_grad()
    def _validate(self, model, loader, device, limit=None):
        model.eval()
        for i, batch in enumerate(loader):
            if limit is not None and i >= limit:
                break
            batch = self._to_device(batch, device)
            out = model.validation_step(batch, i)
            for cb in self.callbacks:
                cb.on_validation_batch_end(self, model, out, batch, i)
        for cb in self.callbacks:
            cb.on_validation_epoch_end(self, model)
        self._flush_metrics()
        model.train()

    def save_checkpoint(self, path: str, model=None):
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        sd = {k: v.detach().cpu().contiguous().clone() for k, v in model.state_dict().items()}
        def plain(x):                      # plain containers only: loadable with torch.load(weights_only=True)
            if isinstance(x, dict):
                return {str(k): plain(v) for k, v in x.items()}
            if isinstance(x, (list, tuple)):
                return [plain(v) for v in x]
            return x if isinstance(x, (int, float, str, bool, type(None))) else str(x)
        ckpt = {"state_dict": sd, "epoch": self.current_epoch, "global_step": self.global_step,
                "hyper_parameters": plain(dict(getattr(model, "hparams", {})))}
        opt = getattr(self, "optimizer", None)
        opts = list(opt) if isinstance(opt, (list, tuple)) else [opt]
        if all(o is not None and hasattr(o, "state_dict") for o in opts):
            ckpt["optimizer_states"] = [o.state_dict() for o in opts]
        torch.save(ckpt, path)
        self.checkpoint_callback.best_model_path = os.path.abspath(path)

    def test(self, *_, **__):
        raise NotImplementedError("the reference DDPM defines no test_step/test_dataloader (SURVEY.md section 5)")
