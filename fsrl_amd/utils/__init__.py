"""Host utilities mirroring fsrl.utils (only what the policy-update path touches)."""
from fsrl_amd.utils.exp_util import seed_all
from fsrl_amd.utils.logger import BaseLogger, DummyLogger
from fsrl_amd.utils.optim_util import LagrangianOptimizer

__all__ = ["BaseLogger", "DummyLogger", "LagrangianOptimizer", "seed_all"]
