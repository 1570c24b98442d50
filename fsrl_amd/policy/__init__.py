"""Policy package: the reference's class names over the HIP engine (fsrl/policy/__init__.py)."""
from fsrl_amd.policy.base_policy import BasePolicy
from fsrl_amd.policy.lagrangian_base import LagrangianPolicy
from fsrl_amd.policy.ppo_lag import PPOLagrangian

__all__ = ["BasePolicy", "LagrangianPolicy", "PPOLagrangian"]
