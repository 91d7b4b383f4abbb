"""TensorBoard-free logger with the slice of TensorBoardLogger's API the path uses:
log_metrics / log_hyperparams and `experiment.add_image` (configs/logger/tensorboard.yaml,
callbacks/visualization.py:24-38).  Scalars go to metrics.jsonl, images to PNG files."""
from __future__ import annotations

import json
import os

import numpy as np


class _Experiment:
    def __init__(self, root):
        self.root = root

    def add_image(self, tag, img, global_step=0):
        from PIL import Image
        arr = img.detach().float().cpu().clamp(0, 1).numpy()
        arr = (arr.transpose(1, 2, 0) * 255 + 0.5).astype(np.uint8)
        if arr.shape[2] == 1:
            arr = arr[:, :, 0]
        path = os.path.join(self.root, "images", tag.replace("/", "_"))
        os.makedirs(path, exist_ok=True)
        Image.fromarray(arr).save(os.path.join(path, f"{int(global_step):06d}.png"))

    def add_scalar(self, tag, value, global_step=0):
        with open(os.path.join(self.root, "metrics.jsonl"), "a") as f:
            f.write(json.dumps({"step": int(global_step), tag: float(value)}) + "\n")


class TensorBoardLogger:
    def __init__(self, save_dir="tensorboard/", name="", version="", **unused):
        self.log_dir = os.path.join(save_dir, str(name or ""), str(version or ""))
        os.makedirs(self.log_dir, exist_ok=True)
        self.experiment = _Experiment(self.log_dir)

    def log_metrics(self, metrics, step=0):
        with open(os.path.join(self.log_dir, "metrics.jsonl"), "a") as f:
            f.write(json.dumps({"step": int(step), **{k: float(v) for k, v in metrics.items()}}) + "\n")

    def log_hyperparams(self, params):
        with open(os.path.join(self.log_dir, "hparams.json"), "w") as f:
            json.dump(params, f, indent=1, default=str)
