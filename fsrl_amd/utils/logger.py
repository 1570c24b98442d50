"""The scalar sink the policy and the trainer write to.

Call surface of fsrl/utils/logger/base_logger.py as this path uses it: `store(tab=None, **scalars)` accumulates running
means between two `write(step)` calls; `stats_mean` / `get_mean` read them; `save_checkpoint(suffix)` pickles whatever
`setup_checkpoint_fn` registered; `progress.txt` receives one tab-separated row per `write`.  TensorBoard / W&B writers
are outside this path; `DummyLogger` drops everything."""
import os
import time
from typing import Callable, Dict, Iterable, List, Optional

import numpy as np


class _Means:
    """name -> running mean, in first-seen order"""

    def __init__(self) -> None:
        self._sum: Dict[str, float] = {}
        self._n: Dict[str, int] = {}

    def add(self, name: str, value) -> None:
        v = value if isinstance(value, (int, float)) else float(np.mean(value))
        self.add_sum(name, v, 1)

    def add_sum(self, name: str, total: float, count: int) -> None:
        self._sum[name] = self._sum.get(name, 0.0) + total
        self._n[name] = self._n.get(name, 0) + count

    def names(self) -> List[str]:
        return list(self._sum)

    def mean(self, name: str) -> float:
        n = self._n.get(name, 0)
        return self._sum[name] / n if n else 0.0


class BaseLogger:
    def __init__(self, log_dir: Optional[str] = None, log_txt: bool = True, name: Optional[str] = None):
        self.name = time.strftime("%Y-%m-%d_exp") if name is None else name
        self.log_dir = None if log_dir is None else os.path.join(log_dir, self.name)
        self.checkpoint_fn: Optional[Callable] = None
        self._table = None                     # progress.txt, opened lazily with its header row
        self._want_table = bool(log_txt and self.log_dir)
        if self.log_dir:
            os.makedirs(self.log_dir, exist_ok=True)
        self.reset_data()

    # ------------------------------------------------------------------ accumulate / read
    def reset_data(self) -> None:
        self._means = _Means()

    def store(self, tab: Optional[str] = None, **kwargs) -> None:
        prefix = "" if tab is None else tab + "/"
        for key, value in kwargs.items():
            self._means.add(prefix + key, value)

    def store_rows(self, keys, rows) -> None:
        """`rows[i][j]` = value of `keys[j]` at optimiser step i: the same running means as one store() per row, taken
        as column sums -- the policies hand over the whole per-update table the device wrote (hundreds of rows)."""
        rows = np.asarray(rows, np.float64)
        if rows.size == 0:
            return
        for key, total in zip(keys, rows.sum(axis=0)):
            self._means.add_sum(key, float(total), rows.shape[0])

    @property
    def logger_keys(self) -> Iterable[str]:
        return self._means.names()

    def get_mean(self, key: str) -> float:
        return self._means.mean(key) if key in self._means.names() else 0.0

    def get_mean_list(self, keys: Iterable[str]) -> List[float]:
        return [self.get_mean(k) for k in keys]

    def get_mean_dict(self, keys: Iterable[str]) -> Dict[str, float]:
        return {k: self.get_mean(k) for k in keys}

    @property
    def stats_mean(self) -> Dict[str, float]:
        return self.get_mean_dict(self.logger_keys)

    # ------------------------------------------------------------------ flush
    def write(self, step: int, display: bool = False, display_keys: Iterable[str] = None) -> None:
        if "update/env_step" not in self.logger_keys:
            self.store(tab="update", env_step=step)
        if self._want_table:
            keys = list(self.logger_keys)
            if self._table is None:
                self._table = open(os.path.join(self.log_dir, "progress.txt"), "w")
                self._table.write("\t".join(["Steps"] + keys) + "\n")
            self._table.write("\t".join(str(v) for v in [step] + self.get_mean_list(keys)) + "\n")
            self._table.flush()
        if display:
            self.display_tabular(display_keys)
        self.reset_data()

    def write_without_reset(self, *args, **kwarg) -> None:
        """hook of the reference's TensorBoard / W&B loggers: nothing to do for the text table"""

    def display_tabular(self, display_keys: Iterable[str] = None) -> None:
        keys = sorted(display_keys or self.logger_keys)
        width = max([15] + [len(k) for k in keys])
        rule = "-" * (width + 22)
        print(rule)
        for k in keys:
            print(f"| {k:>{width}} | {self.get_mean(k):15.5g} |")
        print(rule, flush=True)

    # ------------------------------------------------------------------ checkpoints / config
    def setup_checkpoint_fn(self, checkpoint_fn: Optional[Callable] = None) -> None:
        self.checkpoint_fn = checkpoint_fn

    def save_checkpoint(self, suffix=None) -> None:
        if not (self.checkpoint_fn and self.log_dir):
            return
        import torch
        folder = os.path.join(self.log_dir, "checkpoint")
        os.makedirs(folder, exist_ok=True)
        tag = "" if suffix is None else "_" + (str(suffix) if not isinstance(suffix, int) else "%d" % suffix)
        torch.save(self.checkpoint_fn(), os.path.join(folder, "model" + tag + ".pt"))

    def save_config(self, config: dict, verbose=True) -> None:
        if self.log_dir:
            import yaml
            with open(os.path.join(self.log_dir, "config.yaml"), "w") as f:
                yaml.dump(config, f, default_flow_style=False, indent=4, sort_keys=False)

    def restore_data(self) -> None:
        """nothing persistent to restore for the text table"""

    def print(self, msg: str, color="green") -> None:
        print(msg)


class DummyLogger(BaseLogger):
    """The sink of runs that log nothing: every call is accepted and dropped (no accumulation either -- the policies store
    a dozen scalars per optimiser step), reads return zeros / an empty dict."""

    def __init__(self, *args, **kwarg) -> None:
        self.name, self.log_dir, self.checkpoint_fn = "dummy", None, None
        self._table, self._want_table = None, False
        self.reset_data()

    def store(self, *args, **kwarg) -> None:
        pass

    def store_rows(self, *args, **kwarg) -> None:
        pass

    def write(self, *args, **kwarg) -> None:
        pass

    def print(self, *args, **kwarg) -> None:
        pass

    def save_checkpoint(self, *args, **kwarg) -> None:
        pass

    def get_mean(self, key: str) -> float:
        return 0.0

    @property
    def stats_mean(self) -> dict:
        return {}
