"""A minimal attribute bag with the few `tianshou.data.Batch` behaviours the collector and the
policy interface use (attribute / key access, boolean-mask and slice indexing, update, get)."""
import numpy as np


class Batch:
    def __init__(self, **kwargs):
        self.__dict__.update(kwargs)

    def __getitem__(self, index):
        if isinstance(index, str):
            return self.__dict__[index]
        out = Batch()
        for k, v in self.__dict__.items():
            if isinstance(v, Batch):
                out.__dict__[k] = v[index] if len(v.__dict__) else Batch()
            elif isinstance(v, np.ndarray):
                out.__dict__[k] = v[index]
            elif isinstance(v, dict):
                out.__dict__[k] = {kk: (vv[index] if isinstance(vv, np.ndarray) else vv) for kk, vv in v.items()}
            else:
                out.__dict__[k] = v
        return out

    def __setitem__(self, key, value):
        self.__dict__[key] = value

    def __contains__(self, key):
        return key in self.__dict__

    def get(self, key, default=None):
        return self.__dict__.get(key, default)

    def pop(self, key, default=None):
        return self.__dict__.pop(key, default)

    def update(self, other=None, **kwargs):
        if other is not None:
            kwargs = dict(other.__dict__ if isinstance(other, Batch) else other, **kwargs)
        self.__dict__.update(kwargs)

    def keys(self):
        return self.__dict__.keys()

    def __len__(self):
        for v in self.__dict__.values():
            if isinstance(v, np.ndarray) and v.ndim > 0:
                return len(v)
        return 0

    def __repr__(self):
        return "Batch(" + ", ".join(f"{k}={type(v).__name__}" for k, v in self.__dict__.items()) + ")"
