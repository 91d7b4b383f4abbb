#!/usr/bin/env python3
"""Entry point with the reference's command line:  python run.py experiment=ddpm/cifar10 [key=value ...]

Uses Hydra when it is installed; otherwise the built-in composer (src/runtime/config.py) reads
the same configs/ tree.  Replaces run.py:5-15 of the reference."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)


def _run(config):
    from src.train import train
    from src.utils import utils
    if config.get("print_config"):
        utils.print_config(config, resolve=True)
    return train(config)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    try:
        import hydra                                    # noqa: F401
    except ImportError:
        from src.runtime.config import Composer
        config = Composer(os.path.join(HERE, "configs")).compose("config", argv)
        run_dir = os.path.join(config.get("log_dir", "logs"), "runs", str(config.get("exp_name", "run")))
        os.makedirs(run_dir, exist_ok=True)
        os.chdir(run_dir)                               # hydra.job.chdir=True behaviour (configs/hydra/default.yaml)
        return _run(config)
    import hydra
    return hydra.main(config_path="configs", config_name="config", version_base="1.1")(_run)()


if __name__ == "__main__":
    main()
