"""`VectorReplayBuffer` proxy whose storage is the HIP-resident store of an Engine.

Interface used by the reference: `add(batch, buffer_ids) -> (ptr, ep_rew, ep_len, ep_idx)`
(fsrl/data/fast_collector.py:333-335), `reset(keep_statistics)` (trainer/onpolicy.py:109),
`__len__`, `buffer_num`.  Rows go to pinned staging and reach HBM by hipMemcpyAsync on the
library's side stream (`fsrl_store_push`)."""
import numpy as np


class HipVectorReplayBuffer:
    def __init__(self, engine, total_size=None, buffer_num=None):
        self.engine = engine
        self.buffer_num = engine.cfg.env_num if buffer_num is None else buffer_num
        assert self.buffer_num <= engine.cfg.env_num, "more sub-buffers requested than the engine has"
        self.maxsize = engine.cfg.buffer_size

    def add(self, batch, buffer_ids=None):
        ids = np.arange(len(batch.rew)) if buffer_ids is None else np.asarray(buffer_ids)
        info = batch.get("info", None)
        cost = batch.get("cost", None)
        if cost is None and info is not None:
            cost = info.get("cost", None) if hasattr(info, "get") else None
        if cost is None:
            cost = np.zeros(len(ids))
        return self.engine.push(ids, batch.obs, batch.act, batch.rew, cost, batch.terminated,
                                batch.truncated, batch.obs_next)

    def reset(self, keep_statistics: bool = False) -> None:
        self.engine.reset_store(keep_statistics)

    def __len__(self) -> int:
        return len(self.engine)

    def sample_indices(self, batch_size: int):
        assert batch_size == 0, "only the on-policy sample(0) order is exposed in this round"
        return self.engine.sample0()
