from fsrl_amd.trainer.base_trainer import BaseTrainer
from fsrl_amd.trainer.onpolicy import OnpolicyTrainer

__all__ = ["BaseTrainer", "OnpolicyTrainer"]
