"""`BaseModel` / `ValidationResult` with the reference's surface (src/models/base.py:7-31).

When pytorch_lightning is importable the real LightningModule is the base class, so the model
runs under Lightning's Trainer unchanged; otherwise the small stand-in from
src/runtime/lightning_lite.py provides the hooks the DDPM path uses
(save_hyperparameters / hparams / log / device).
"""
from dataclasses import dataclass, field
from typing import Optional

import torch

try:                                                    # pragma: no cover - not installed in this image
    from pytorch_lightning import LightningModule
except Exception:                                       # noqa: BLE001
    from ..runtime.lightning_lite import LightningModule


@dataclass
class ValidationResult:
    others: dict = field(default_factory=dict)
    real_image: Optional[torch.Tensor] = None
    fake_image: Optional[torch.Tensor] = None
    recon_image: Optional[torch.Tensor] = None
    label: Optional[torch.Tensor] = None
    encode_latent: Optional[torch.Tensor] = None


def _cfg(obj, name):
    return obj[name] if isinstance(obj, dict) else getattr(obj, name)


class BaseModel(LightningModule):
    """Reads image geometry from the *datamodule config* (not the object), like the reference."""

    def __init__(self, datamodule) -> None:
        super().__init__()
        self.width = _cfg(datamodule, "width")
        self.height = _cfg(datamodule, "height")
        self.channels = _cfg(datamodule, "channels")
        self.input_normalize = _cfg(_cfg(datamodule, "transforms"), "normalize")
        self.output_act = "tanh" if self.input_normalize else "sigmoid"
