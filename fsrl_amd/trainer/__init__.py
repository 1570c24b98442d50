from fsrl_amd.trainer.base_trainer import BaseTrainer
from fsrl_amd.trainer.onpolicy import OnpolicyTrainer
from fsrl_amd.trainer.offpolicy import OffpolicyTrainer

__all__ = ["BaseTrainer", "OnpolicyTrainer", "OffpolicyTrainer"]
