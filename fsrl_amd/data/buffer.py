"""`VectorReplayBuffer` proxy whose storage is the HIP-resident store of an Engine.

Interface used by the reference: `add(batch, buffer_ids) -> (ptr, ep_rew, ep_len, ep_idx)`
(fsrl/data/fast_collector.py:333-335), `reset(keep_statistics)` (trainer/onpolicy.py:109),
`__len__`, `buffer_num`.  Rows go to pinned staging and reach HBM by hipMemcpyAsync on the
library's side stream (`fsrl_store_push`)."""
import numpy as np


class HipVectorReplayBuffer:
    def __init__(self, engine, total_size=None, buffer_num=None):
        self.engine = engine
        self.buffer_num = engine.cfg.env_num if buffer_num is None else int(buffer_num)
        assert self.buffer_num <= engine.cfg.env_num, "more sub-buffers requested than the engine has"
        total = engine.cfg.buffer_size if total_size is None else int(total_size)
        # the reference cuts VectorReplayBuffer(total_size, buffer_num) into ceil(total / num)-row sub-buffers
        # (base_agent.py:279): take exactly that geometry (slot ids, wrap-around, sample indices depend on it);
        # it has to fit what the engine allocated, otherwise the C ABI refuses (AssertionError)
        want_sub = -(-total // self.buffer_num)
        if (want_sub, self.buffer_num) != engine.store_geometry():
            engine.store_configure(total, self.buffer_num)
        self._sub, _ = engine.store_geometry()
        self.maxsize = self._sub * self.buffer_num
        self._sizes = np.zeros(engine.cfg.env_num, np.int64)             # host mirror of the fill levels

    def add(self, batch, buffer_ids=None):
        ids = np.arange(len(batch.rew)) if buffer_ids is None else np.asarray(buffer_ids)
        info = batch.get("info", None)
        cost = batch.get("cost", None)
        if cost is None and info is not None:
            cost = info.get("cost", None) if hasattr(info, "get") else None
        if cost is None:
            cost = np.zeros(len(ids))
        for e in ids:
            self._sizes[e] = min(self._sizes[e] + 1, self._sub)
        return self.engine.push(ids, batch.obs, batch.act, batch.rew, cost, batch.terminated,
                                batch.truncated, batch.obs_next)

    def note_add(self, buffer_ids) -> None:
        """Bookkeeping for rows pushed directly through the engine (FastCollector device_actor path)."""
        for e in buffer_ids:
            self._sizes[e] = min(self._sizes[e] + 1, self._sub)

    def sync_sizes(self) -> None:
        """Fill levels from the engine (rows stored through fsrl_collect_step bypass this proxy)."""
        self._sizes[:] = self.engine.store_sizes()

    def reset(self, keep_statistics: bool = False) -> None:
        self._sizes[:] = 0
        self.engine.reset_store(keep_statistics)

    def __len__(self) -> int:
        return len(self.engine)

    def sample_indices(self, batch_size: int):
        """tianshou-0.5 ReplayBufferManager.sample_indices: batch_size == 0 -> every row, env-major
        and chronological; > 0 -> sub-buffers drawn proportionally to their fill with the numpy
        global RNG, then uniform rows inside each (np.random.choice), concatenated in buffer order."""
        if batch_size == 0:
            return self.engine.sample0()
        if batch_size < 0:
            return np.array([], int)
        lengths = self._sizes[:self.buffer_num]
        pick = np.random.choice(self.buffer_num, batch_size, p=lengths / lengths.sum())
        counts = np.bincount(pick, minlength=self.buffer_num)
        return np.concatenate([np.random.choice(int(lengths[e]), int(counts[e])) + e * self._sub
                               for e in range(self.buffer_num)]).astype(np.int64)
