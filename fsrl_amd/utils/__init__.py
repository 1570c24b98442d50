"""Host utilities of the policy-update path: seeding, the scalar logger, the PID multiplier."""
from fsrl_amd._lazy import install

install(__name__, globals(), {
    "seed_all": "exp_util",
    "BaseLogger": "logger",
    "DummyLogger": "logger",
    "LagrangianOptimizer": "optim_util",
})
