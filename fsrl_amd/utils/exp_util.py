"""seed_all: the three RNGs the reference seeds (fsrl/utils/exp_util.py:16-30)."""
import os
import random

import numpy as np
import torch


def seed_all(seed=1029, others=None):
    random.seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if others is not None:
        for item in (others if hasattr(others, "__iter__") else [others]):
            if hasattr(item, "seed"):
                item.seed(seed)
