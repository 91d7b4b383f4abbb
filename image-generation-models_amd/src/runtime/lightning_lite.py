"""Minimal stand-ins for the pytorch_lightning classes the DDPM path touches (SURVEY.md
App. E): LightningModule (save_hyperparameters / hparams / log / device), Callback,
LightningDataModule, seed_everything.  Only used when pytorch_lightning is absent."""
from __future__ import annotations

import inspect
import random
from typing import Any, Dict

import numpy as np
import torch
from torch import nn


class AttrDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


class LightningModule(nn.Module):
    def __init__(self):
        super().__init__()
        self._hparams = AttrDict()
        self._logged: Dict[str, Any] = {}
        self.trainer = None

    def save_hyperparameters(self, *_, **__):
        """Capture the calling __init__'s arguments (what Lightning does by frame inspection)."""
        frame = inspect.currentframe().f_back
        args = inspect.getargvalues(frame)
        hp = {k: args.locals[k] for k in args.args if k not in ("self",)}
        if args.keywords and args.keywords in args.locals:
            hp.update(args.locals[args.keywords])
        hp.pop("__class__", None)
        self._hparams.update(hp)

    @property
    def hparams(self):
        return self._hparams

    @property
    def device(self):
        for p in self.parameters():
            return p.device
        return torch.device("cpu")

    # manual optimization (automatic_optimization = False): the two calls the reference's GAN models use
    automatic_optimization = True

    def optimizers(self):
        opts = getattr(self, "_optimizers", None)
        if opts is None:
            cfg = self.configure_optimizers()
            opts = list(cfg) if isinstance(cfg, (list, tuple)) else [cfg]
            object.__setattr__(self, "_optimizers", opts)
        return opts if len(opts) > 1 else opts[0]

    def manual_backward(self, loss, *args, **kwargs):
        loss.backward(*args, **kwargs)

    def log(self, name, value, **_):
        self._logged[name] = value
        if self.trainer is not None:
            self.trainer._log_metric(name, value)


class Callback:
    def on_validation_batch_end(self, trainer, pl_module, outputs, batch, batch_idx, dataloader_idx=0):
        pass

    def on_validation_epoch_end(self, trainer, pl_module):
        pass

    def on_train_epoch_end(self, trainer, pl_module):
        pass


class LightningDataModule:
    def prepare_data(self):
        pass

    def setup(self, stage=None):
        pass


def seed_everything(seed: int):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    return seed
