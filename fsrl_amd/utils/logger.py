"""Scalar sink with the interface the policy / trainer call (fsrl/utils/logger/base_logger.py:
store(tab, **scalars), write, print, save_checkpoint, stats_mean, get_mean).  TensorBoard / W&B
writers are out of scope; `progress.txt` TSV output is kept."""
import os
import os.path as osp
import time
from collections import defaultdict
from typing import Callable, Iterable, Optional

import numpy as np


class _RunningMean:
    __slots__ = ("total", "count")

    def __init__(self):
        self.total, self.count = 0.0, 0

    def add(self, v):
        self.total += float(v)
        self.count += 1

    @property
    def mean(self):
        return self.total / self.count if self.count else 0.0


class BaseLogger:
    def __init__(self, log_dir: Optional[str] = None, log_txt: bool = True, name: Optional[str] = None):
        self.name = name if name is not None else time.strftime("%Y-%m-%d_exp")
        self.log_dir = osp.join(log_dir, self.name) if log_dir is not None else None
        self.output_file = None
        if self.log_dir:
            os.makedirs(self.log_dir, exist_ok=True)
            if log_txt:
                self.output_file = open(osp.join(self.log_dir, "progress.txt"), "w")
        self.first_row = True
        self.checkpoint_fn: Optional[Callable] = None
        self.reset_data()

    def setup_checkpoint_fn(self, checkpoint_fn: Optional[Callable] = None) -> None:
        self.checkpoint_fn = checkpoint_fn

    def reset_data(self) -> None:
        self.log_data = defaultdict(_RunningMean)

    def store(self, tab: Optional[str] = None, **kwargs) -> None:
        for k, v in kwargs.items():
            self.log_data[(tab + "/" + k) if tab is not None else k].add(np.mean(v))

    @property
    def logger_keys(self) -> Iterable[str]:
        return self.log_data.keys()

    def get_mean(self, key: str) -> float:
        return self.log_data[key].mean

    def get_mean_list(self, keys: Iterable[str]):
        return [self.get_mean(k) for k in keys]

    def get_mean_dict(self, keys: Iterable[str]):
        return {k: self.get_mean(k) for k in keys}

    @property
    def stats_mean(self) -> dict:
        return self.get_mean_dict(self.logger_keys)

    def write(self, step: int, display: bool = False, display_keys: Iterable[str] = None) -> None:
        if "update/env_step" not in self.logger_keys:
            self.store(tab="update", env_step=step)
        if self.output_file is not None:
            if self.first_row:
                self.output_file.write("\t".join(["Steps"] + list(self.logger_keys)) + "\n")
            vals = [step] + self.get_mean_list(self.logger_keys)
            self.output_file.write("\t".join(map(str, vals)) + "\n")
            self.output_file.flush()
            self.first_row = False
        if display:
            self.display_tabular(display_keys)
        self.reset_data()

    def write_without_reset(self, *args, **kwarg) -> None:
        pass

    def display_tabular(self, display_keys: Iterable[str] = None) -> None:
        keys = sorted(display_keys or self.logger_keys)
        width = max([15] + [len(k) for k in keys])
        print("-" * (width + 22))
        for k in keys:
            print(f"| {k:>{width}} | {self.get_mean(k):15.5g} |")
        print("-" * (width + 22), flush=True)

    def save_checkpoint(self, suffix=None) -> None:
        if self.checkpoint_fn and self.log_dir:
            import torch
            fpath = osp.join(self.log_dir, "checkpoint")
            os.makedirs(fpath, exist_ok=True)
            suffix = "%d" % suffix if isinstance(suffix, int) else suffix
            fname = "model" + ("_" + suffix if suffix is not None else "") + ".pt"
            torch.save(self.checkpoint_fn(), osp.join(fpath, fname))

    def save_config(self, config: dict, verbose=True) -> None:
        if self.log_dir:
            import yaml
            with open(osp.join(self.log_dir, "config.yaml"), "w") as f:
                yaml.dump(config, f, default_flow_style=False, indent=4, sort_keys=False)

    def restore_data(self) -> None:
        pass

    def print(self, msg: str, color="green") -> None:
        print(msg)


class DummyLogger(BaseLogger):
    """Swallows everything (fsrl/utils/logger/base_logger.py DummyLogger)."""

    def __init__(self, *args, **kwarg) -> None:
        self.reset_data()
        self.checkpoint_fn = None
        self.log_dir = None
        self.output_file = None

    def store(self, *args, **kwarg):
        pass

    def write(self, *args, **kwarg):
        pass

    def print(self, *args, **kwarg):
        pass

    def save_checkpoint(self, *args, **kwarg):
        pass

    def get_mean(self, key):
        return 0.0

    @property
    def stats_mean(self):
        return {}
