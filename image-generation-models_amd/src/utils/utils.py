"""Rank-zero logger, config pretty-printer and hyper-parameter logging
(reference: src/utils/utils.py:13-32, 79-118, 125-166)."""
import logging
import os
from typing import Sequence


def _is_rank_zero() -> bool:
    return int(os.environ.get("RANK", "0")) == 0


def rank_zero_only(fn):
    def wrapped(*a, **k):
        if _is_rank_zero():
            return fn(*a, **k)
        return None
    return wrapped


def get_logger(name=__name__, level=logging.INFO) -> logging.Logger:
    logger = logging.getLogger(name)
    logger.setLevel(level)
    if not logging.getLogger().handlers:
        logging.basicConfig(format="[%(asctime)s][%(name)s][%(levelname)s] - %(message)s")
    for lvl in ("debug", "info", "warning", "error", "exception", "fatal", "critical"):
        setattr(logger, lvl, rank_zero_only(getattr(logger, lvl)))
    return logger


@rank_zero_only
def print_config(config, fields: Sequence[str] = ("trainer", "model", "datamodule", "callbacks", "logger", "seed", "exp_name"),
                 resolve: bool = True) -> None:
    """Print the composed config as a tree (Rich when available) and save it to config_tree.txt."""
    import yaml
    lines = []
    try:
        import rich.syntax
        import rich.tree
        tree = rich.tree.Tree("CONFIG")
        for f in fields:
            if f in config:
                sec = config[f]
                text = yaml.safe_dump(_plain(sec), sort_keys=False) if isinstance(sec, dict) else str(sec)
                tree.add(f).add(rich.syntax.Syntax(text, "yaml"))
                lines.append(f"{f}:\n{text}")
        rich.print(tree)
    except ImportError:
        for f in fields:
            if f in config:
                lines.append(f"{f}:\n{yaml.safe_dump(_plain(config[f]), sort_keys=False)}")
        print("\n".join(lines))
    with open("config_tree.txt", "w") as fp:
        fp.write("\n".join(lines))


def _plain(x):
    if isinstance(x, dict):
        return {k: _plain(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_plain(v) for v in x]
    return x


@rank_zero_only
def log_hyperparameters(config, model, datamodule, trainer, callbacks, logger) -> None:
    """Send the main config sections and parameter counts to the logger."""
    hp = {"trainer": _plain(config["trainer"]), "model": _plain(config["model"]), "datamodule": _plain(config["datamodule"])}
    if "seed" in config:
        hp["seed"] = config["seed"]
    if "callbacks" in config:
        hp["callbacks"] = _plain(config["callbacks"])
    params = list(model.parameters())
    hp["model/params_total"] = sum(p.numel() for p in params)
    hp["model/params_trainable"] = sum(p.numel() for p in params if p.requires_grad)
    hp["model/params_not_trainable"] = sum(p.numel() for p in params if not p.requires_grad)
    if logger is not None and hasattr(logger, "log_hyperparams"):
        logger.log_hyperparams(hp)
