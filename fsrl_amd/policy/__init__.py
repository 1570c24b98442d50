"""The reference's policy class names over the HIP engine (what fsrl.policy exports, one per algorithm)."""
from fsrl_amd._lazy import install

install(__name__, globals(), {
    "BasePolicy": "base_policy",
    "LagrangianPolicy": "lagrangian_base",
    "PPOLagrangian": "ppo_lag",
    "TRPOLagrangian": "trpo_lag",
    "CPO": "cpo",
    "FOCOPS": "focops",
    "SACLagrangian": "sac_lag",
    "DDPGLagrangian": "ddpg_lag",
    "CVPO": "cvpo",
    "PolicyGroup": "grouped",
})
