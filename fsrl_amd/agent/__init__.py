"""The seven agents of the reference (fsrl.agent) with their keyword arguments, over the HIP-backed policies."""
from fsrl_amd._lazy import install

install(__name__, globals(), {
    "BaseAgent": "base_agent",
    "OnpolicyAgent": "base_agent",
    "OffpolicyAgent": "sac_lag_agent",
    "PPOLagAgent": "ppo_lag_agent",
    "CPOAgent": "trust_agents",
    "TRPOLagAgent": "trust_agents",
    "FOCOPSAgent": "trust_agents",
    "SACLagAgent": "sac_lag_agent",
    "DDPGLagAgent": "ddpg_lag_agent",
    "CVPOAgent": "cvpo_agent",
})
