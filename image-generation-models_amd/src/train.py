"""`train(config)`: builds datamodule, model, callbacks, logger and trainer from the composed
config and fits (reference: src/train.py:18-79).  Differences that the AMD box forces:
the nvidia-smi based GPU picker (GPUtil, train.py:44-45) is replaced by LOCAL_RANK / device 0,
and hydra/lightning fall back to the in-tree stand-ins when they are not installed."""
from typing import List, Optional

from src.utils import utils

try:
    from hydra.utils import instantiate
except ImportError:
    from src.runtime.config import instantiate

try:
    from pytorch_lightning import seed_everything
except ImportError:
    from src.runtime.lightning_lite import seed_everything

log = utils.get_logger(__name__)


def train(config) -> Optional[float]:
    if "seed" in config:
        seed_everything(config.seed)

    log.info(f"Instantiating datamodule <{config.datamodule._target_}>")
    datamodule = instantiate(config.datamodule)

    log.info(f"Instantiating model <{config.model._target_}>")
    model = instantiate(config.model, datamodule=config.datamodule, _recursive_=False)

    callbacks: List = []
    for _, cb_conf in (config.get("callbacks") or {}).items():
        if isinstance(cb_conf, dict) and "_target_" in cb_conf:
            log.info(f"Instantiating callback <{cb_conf._target_}>")
            callbacks.append(instantiate(cb_conf))

    log.info(f"Instantiating logger <{config.logger._target_}>")
    logger = instantiate(config.logger)

    log.info(f"Instantiating trainer <{config.trainer._target_}>")
    trainer = instantiate(config.trainer, callbacks=callbacks, logger=logger, _convert_="partial")

    log.info("Logging hyperparameters!")
    utils.log_hyperparameters(config=config, model=model, datamodule=datamodule, trainer=trainer,
                              callbacks=callbacks, logger=logger)

    log.info("Starting training!")
    trainer.fit(model=model, datamodule=datamodule)

    if config.get("test_after_training") and not config.trainer.get("fast_dev_run"):
        log.info("Starting testing!")
        trainer.test()

    log.info("Finalizing!")
    ckpt = getattr(getattr(trainer, "checkpoint_callback", None), "best_model_path", None)
    log.info(f"Best checkpoint path:\n{ckpt}")

    metric = config.get("optimized_metric")
    if metric:
        return trainer.callback_metrics[metric]
    return None
