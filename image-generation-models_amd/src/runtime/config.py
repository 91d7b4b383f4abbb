"""A small Hydra-subset composer, used when hydra-core/omegaconf are not installed.

Covers exactly what the DDPM entry needs: a `defaults` list with config groups,
`_self_`, `override /group: name`, `override /group@_global_: name`, `# @package _global_`
files, `${a.b}` / `${now:%fmt}` / `${hydra:runtime.cwd}` interpolation, command-line
`a.b=c`, `+a.b=c`, `group=name`, and `_target_` instantiation (with `_recursive_=False` semantics:
nested `_target_` dicts are passed through as configs).
"""
from __future__ import annotations

import datetime
import importlib
import os
import re
from typing import Any, Dict, List, Optional, Tuple

import yaml


class Cfg(dict):
    """dict with attribute access (the part of DictConfig the path uses)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def get(self, k, default=None):
        return dict.get(self, k, default)


def _wrap(x):
    if isinstance(x, dict):
        return Cfg({k: _wrap(v) for k, v in x.items()})
    if isinstance(x, list):
        return [_wrap(v) for v in x]
    return x


def _merge(dst: dict, src: dict):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = v


class _Loader(yaml.SafeLoader):
    """YAML 1.1 reads `1e-4` as a string (a float needs a dot there); OmegaConf, which the reference's configs are written
    for, reads it as a float.  Same implicit resolver here so `lrG: 1e-4` (configs/model/wgan_gp.yaml) is a number."""


_Loader.add_implicit_resolver(
    "tag:yaml.org,2002:float",
    re.compile(r"""^(?:[-+]?(?:[0-9][0-9_]*)\.[0-9_]*(?:[eE][-+]?[0-9]+)?
                    |[-+]?(?:[0-9][0-9_]*)(?:[eE][-+]?[0-9]+)
                    |\.[0-9_]+(?:[eE][-+][0-9]+)?
                    |[-+]?\.(?:inf|Inf|INF)
                    |\.(?:nan|NaN|NAN))$""", re.X),
    list("-+0123456789."))


def _yaml(text: str):
    return yaml.load(text, Loader=_Loader)


def _load(path: str) -> Tuple[dict, bool]:
    text = open(path).read()
    is_global = bool(re.match(r"\s*#\s*@package\s+_global_", text))
    return (_yaml(text) or {}), is_global


def _parse_value(s: str):
    try:
        return _yaml(s)
    except yaml.YAMLError:
        return s


class Composer:
    def __init__(self, config_dir: str):
        self.dir = config_dir

    def _group_file(self, group: str, name: str) -> str:
        return os.path.join(self.dir, group, name + ".yaml")

    def _include(self, out: dict, group: str, name: Optional[str], package: Optional[str], choices: Dict[str, str]):
        """Merge config group file `group/name` into out (under `group` unless global)."""
        if name is None or name == "null":
            return
        body, is_global = _load(self._group_file(group, name))
        defaults = body.pop("defaults", [])
        pending_self = True
        for d in defaults:
            if d == "_self_":
                self._place(out, body, group, package, is_global); pending_self = False
                continue
            if isinstance(d, str):                       # sibling file in the same group
                self._include(out, group, d, package, choices)
                continue
            (key, val), = d.items()
            if key.startswith("override "):              # recorded by compose() in its first pass
                continue
            g = key.lstrip("/")
            g, _, pkg = g.partition("@")
            self._include(out, g, choices.get(g, val), pkg or None, choices)
        if pending_self:
            self._place(out, body, group, package, is_global)

    @staticmethod
    def _place(out: dict, body: dict, group: str, package: Optional[str], is_global: bool):
        if is_global or package == "_global_":
            _merge(out, body)
        else:
            node = out
            for part in (package or group).split("/"):
                node = node.setdefault(part, {})
            _merge(node, body)

    def _collect_overrides(self, group: str, name: Optional[str], choices: Dict[str, str], packages: Dict[str, str]):
        if name is None or name == "null" or not os.path.exists(self._group_file(group, name)):
            return
        body, _ = _load(self._group_file(group, name))
        for d in body.get("defaults", []):
            if isinstance(d, dict):
                (key, val), = d.items()
                if key.startswith("override "):
                    g = key[len("override "):].lstrip("/")
                    g, _, pkg = g.partition("@")
                    choices[g] = val          # `@_global_` on an override names the parent's package, not a new
                                              # location: the group keeps its own package (callbacks -> config.callbacks)
                    self._collect_overrides(g, val, choices, packages)

    def compose(self, config_name: str = "config", overrides: Optional[List[str]] = None) -> Cfg:
        overrides = list(overrides or [])
        root, _ = _load(os.path.join(self.dir, config_name + ".yaml"))
        defaults = root.pop("defaults", [])
        groups = {}
        order: List[Tuple[str, Optional[str]]] = []
        for d in defaults:
            if d == "_self_":
                order.append(("_self_", None)); continue
            (key, val), = d.items()
            if key.startswith("override "):
                continue
            groups[key] = val
            order.append((key, val))
        dotted = []
        for ov in overrides:                              # group selections from the command line
            k, _, v = ov.partition("=")
            if k.lstrip("+") in groups and "." not in k:
                groups[k.lstrip("+")] = None if v in ("null", "") else v
            else:
                dotted.append((k, v))
        choices = dict(groups)
        packages: Dict[str, str] = {}
        for g in list(groups):                            # `override /x: y` inside selected files (experiment, model)
            self._collect_overrides(g, choices.get(g), choices, packages)
        out: dict = {}
        for key, _ in order:
            if key == "_self_":
                _merge(out, root)
            elif key != "hydra":
                self._include(out, key, choices.get(key), packages.get(key), choices)
        for k, v in dotted:
            node = out
            parts = k.lstrip("+").split(".")
            for p in parts[:-1]:
                node = node.setdefault(p, {})
            node[parts[-1]] = _parse_value(v)
        out.pop("hydra", None)
        cfg = _wrap(out)
        _resolve(cfg, cfg)
        return cfg


_INTERP = re.compile(r"\$\{([^{}]+)\}")
_NOW = datetime.datetime.now()


def _lookup(root: dict, path: str):
    node = root
    for p in path.split("."):
        node = node[p]
    return node


def _resolve_str(s: str, root: dict):
    for _ in range(10):
        m = _INTERP.search(s)
        if not m:
            break
        expr = m.group(1)
        if expr.startswith("now:"):
            val = _NOW.strftime(expr[4:])
        elif expr.startswith("hydra:"):
            val = os.getcwd() if expr == "hydra:runtime.cwd" else ""
        elif expr.startswith("oc.env:"):
            val = os.environ.get(expr[7:].split(",")[0], "")
        else:
            val = _lookup(root, expr)
        if m.span() == (0, len(s)) and not isinstance(val, str):
            return val
        s = s[:m.start()] + str(val) + s[m.end():]
    return s


def _resolve(node, root):
    it = node.items() if isinstance(node, dict) else enumerate(node)
    for k, v in list(it):
        if isinstance(v, str) and "${" in v:
            node[k] = _resolve_str(v, root)
        elif isinstance(v, (dict, list)):
            _resolve(v, root)


# ---- _target_ instantiation ------------------------------------------------------------------
# Targets that name pytorch_lightning classes resolve to the in-tree stand-ins when Lightning is absent.
_ALIASES = {
    "pytorch_lightning.Trainer": "src.runtime.trainer.Trainer",
    "pytorch_lightning.loggers.tensorboard.TensorBoardLogger": "src.runtime.loggers.TensorBoardLogger",
    "pytorch_lightning.callbacks.progress.TQDMProgressBar": "src.runtime.trainer.ProgressBar",
}


_ROOT = __name__.rsplit(".src.runtime", 1)[0] if ".src.runtime" in __name__ else ""


def locate(path: str):
    if _ROOT and path.startswith("src."):          # package imported under a parent name (tests): keep ONE copy of each module
        path = _ROOT + "." + path
    try:
        mod, _, name = path.rpartition(".")
        return getattr(importlib.import_module(mod), name)
    except (ImportError, AttributeError):
        if path in _ALIASES:
            return locate(_ALIASES[path])
        raise


def instantiate(cfg, *args, **kwargs):
    """hydra.utils.instantiate(cfg, ..., _recursive_=False) for a dict with `_target_`."""
    kwargs.pop("_recursive_", None); kwargs.pop("_convert_", None)
    params = {k: v for k, v in cfg.items() if not k.startswith("_")}
    params.update(kwargs)
    return locate(cfg["_target_"])(*args, **params)
