from fsrl_amd.agent.base_agent import BaseAgent, OnpolicyAgent
from fsrl_amd.agent.ppo_lag_agent import PPOLagAgent

__all__ = ["BaseAgent", "OnpolicyAgent", "PPOLagAgent"]
