from fsrl_amd.env.shmem import ShmemVectorEnv
from fsrl_amd.env.synthetic import Box, SyntheticSafetyVectorEnv

__all__ = ["Box", "ShmemVectorEnv", "SyntheticSafetyVectorEnv"]
