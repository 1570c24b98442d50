"""Policy package: the reference's class names over the HIP engine (fsrl/policy/__init__.py)."""
from fsrl_amd.policy.base_policy import BasePolicy
from fsrl_amd.policy.lagrangian_base import LagrangianPolicy
from fsrl_amd.policy.ppo_lag import PPOLagrangian
from fsrl_amd.policy.trpo_lag import TRPOLagrangian
from fsrl_amd.policy.cpo import CPO
from fsrl_amd.policy.sac_lag import SACLagrangian
from fsrl_amd.policy.ddpg_lag import DDPGLagrangian
from fsrl_amd.policy.focops import FOCOPS
from fsrl_amd.policy.cvpo import CVPO

__all__ = ["BasePolicy", "LagrangianPolicy", "PPOLagrangian", "TRPOLagrangian", "CPO", "SACLagrangian", "DDPGLagrangian", "FOCOPS", "CVPO"]
