from fsrl_amd.env.shmem import ShmemVectorEnv
from fsrl_amd.env.synthetic import Box, SyntheticSafetyVectorEnv
from fsrl_amd.env.toy import PointCircleEnv
from fsrl_amd.env.venv import DummyVectorEnv, EnvList

__all__ = ["Box", "DummyVectorEnv", "EnvList", "PointCircleEnv", "ShmemVectorEnv", "SyntheticSafetyVectorEnv"]
