from fsrl_amd.agent.base_agent import BaseAgent, OnpolicyAgent
from fsrl_amd.agent.ppo_lag_agent import PPOLagAgent
from fsrl_amd.agent.trust_agents import CPOAgent, FOCOPSAgent, TRPOLagAgent
from fsrl_amd.agent.sac_lag_agent import OffpolicyAgent, SACLagAgent
from fsrl_amd.agent.ddpg_lag_agent import DDPGLagAgent
from fsrl_amd.agent.cvpo_agent import CVPOAgent

__all__ = ["BaseAgent", "OnpolicyAgent", "PPOLagAgent", "CPOAgent", "TRPOLagAgent", "OffpolicyAgent", "SACLagAgent", "DDPGLagAgent", "FOCOPSAgent", "CVPOAgent"]
