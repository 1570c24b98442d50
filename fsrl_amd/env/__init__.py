from fsrl_amd.env.synthetic import Box, SyntheticSafetyVectorEnv

__all__ = ["Box", "SyntheticSafetyVectorEnv"]
