"""Stand-ins that let the UNMODIFIED reference (`/root/reference/fsrl/policy/*`) import in
the build container, where tianshou / numba / gymnasium / wandb / tensorboard are absent.

TEST TOOLING ONLY.  Used by `tests/golden/gen_golden.py` (in the build container, never on
the GPU box) to emit golden vectors.  Nothing in the product (`fsrl_amd/`) imports this.

What is restated here is NOT FSRL code, it is the part of the third-party dependency
**tianshou ~= 0.5.0** (reference `setup.py:17`, not vendored, not installed) that FSRL's
policy-update path relies on, written from the published behaviour of that release:

* `Batch` (attribute bag, `__getitem__` by key / index, `split(size, shuffle, merge_last)`)
* `to_numpy`, `to_torch_as`
* `MLP`, `Net`, `ActorProb`, `Actor`, `Critic`   (tianshou.utils.net.*)
* `ReplayBuffer` / `VectorReplayBuffer` index semantics (`sample_indices(0)`, `next`,
  `unfinished_index`, per-env sub-buffers of ceil(total/n) rows)
* `RunningMeanStd`

and `numba.njit`, emulated as "call the python function with python scalars promoted to
numpy float64/int64", which is numba's typing rule (an `array(float32) * float64` is a
float64 multiply in numba, while numpy-2 weak-scalar promotion would keep float32).

Since the tianshou source is not available here, parity of these pieces is "unpinned"
against tianshou itself (DESIGN.md says so); it is pinned against FSRL's call sites.
"""
import math
import sys
import types

import numpy as np
import torch
from torch import nn

SIGMA_MIN, SIGMA_MAX = -20, 2


# ----------------------------------------------------------------------------- Batch
class Batch:
    """Minimal tianshou.data.Batch: an attribute bag of arrays/tensors/sub-batches."""

    def __init__(self, batch_dict=None, **kwargs):
        if batch_dict is not None:
            kwargs = dict(batch_dict, **kwargs)
        for k, v in kwargs.items():
            self.__dict__[k] = self._wrap(v)

    @staticmethod
    def _wrap(v):
        if isinstance(v, dict):
            return Batch(v)
        if isinstance(v, tuple) and len(v) and all(torch.is_tensor(x) for x in v):
            return torch.stack(v)  # Batch(logits=(mu, sigma)) stacks (cpo.py:140-141)
        return v

    def __setattr__(self, k, v):
        self.__dict__[k] = self._wrap(v)

    def __getattr__(self, k):  # only called when missing
        raise AttributeError(k)

    def __contains__(self, k):
        return k in self.__dict__

    def keys(self):
        return self.__dict__.keys()

    def get(self, k, d=None):
        return self.__dict__.get(k, d)

    def pop(self, k, d=None):
        return self.__dict__.pop(k, d)

    def update(self, other=None, **kw):
        if other is not None:
            kw = dict(other.__dict__ if isinstance(other, Batch) else other, **kw)
        for k, v in kw.items():
            setattr(self, k, v)

    def is_empty(self):
        return len(self.__dict__) == 0

    def __len__(self):
        for v in self.__dict__.values():
            if isinstance(v, Batch):
                if not v.is_empty():
                    return len(v)
            elif hasattr(v, "__len__") and not isinstance(v, str):
                return len(v)
        raise TypeError("empty Batch has no len")

    def __getitem__(self, index):
        if isinstance(index, str):
            return self.__dict__[index]
        out = Batch()
        for k, v in self.__dict__.items():
            if isinstance(v, Batch):
                out.__dict__[k] = v[index] if not v.is_empty() else Batch()
            elif v is None or isinstance(v, (torch.distributions.Distribution, )):
                out.__dict__[k] = v
            else:
                out.__dict__[k] = v[index]  # fancy index => copy (numpy and torch)
        return out

    def __setitem__(self, index, value):
        if isinstance(index, str):
            setattr(self, index, value)
            return
        for k, v in value.__dict__.items():
            if isinstance(v, Batch):
                self.__dict__[k][index] = v
            else:
                self.__dict__[k][index] = v

    def split(self, size, shuffle=True, merge_last=False):
        length = len(self)
        if size == -1:
            size = length
        assert size >= 1
        indices = np.random.permutation(length) if shuffle else np.arange(length)
        merge_last = merge_last and length % size > 0
        for idx in range(0, length, size):
            if merge_last and idx + size + size >= length:
                yield self[indices[idx:]]
                break
            yield self[indices[idx:idx + size]]


def to_numpy(x):
    if torch.is_tensor(x):
        return x.detach().cpu().numpy()
    if isinstance(x, Batch):
        return Batch({k: to_numpy(v) for k, v in x.__dict__.items()})
    return np.asanyarray(x)


def to_torch_as(x, y):
    if torch.is_tensor(x):
        return x.to(dtype=y.dtype, device=y.device)
    return torch.from_numpy(np.asanyarray(x)).to(dtype=y.dtype, device=y.device)


class RunningMeanStd:
    def __init__(self, mean=0.0, std=1.0, clip_max=10.0, epsilon=np.finfo(np.float32).eps.item()):
        self.mean, self.var = mean, std
        self.clip_max = clip_max
        self.count = 0
        self.eps = epsilon

    def update(self, data_array):
        batch_mean, batch_var = np.mean(data_array, axis=0), np.var(data_array, axis=0)
        batch_count = len(data_array)
        delta = batch_mean - self.mean
        total_count = self.count + batch_count
        new_mean = self.mean + delta * batch_count / total_count
        m_a = self.var * self.count
        m_b = batch_var * batch_count
        m_2 = m_a + m_b + delta**2 * self.count * batch_count / total_count
        self.mean, self.var, self.count = new_mean, m_2 / total_count, total_count


# ----------------------------------------------------------------------------- nets
class MLP(nn.Module):
    def __init__(self, input_dim, output_dim=0, hidden_sizes=(), norm_layer=None,
                 activation=nn.ReLU, device=None, linear_layer=nn.Linear,
                 flatten_input=True):
        super().__init__()
        self.device = device
        hidden_sizes = [input_dim] + list(hidden_sizes)
        model = []
        for i, o in zip(hidden_sizes[:-1], hidden_sizes[1:]):
            model += [linear_layer(i, o), activation()]
        if output_dim > 0:
            model += [linear_layer(hidden_sizes[-1], output_dim)]
        self.output_dim = output_dim or hidden_sizes[-1]
        self.model = nn.Sequential(*model)
        self.flatten_input = flatten_input

    def forward(self, obs):
        obs = torch.as_tensor(obs, device=self.device, dtype=torch.float32)
        if self.flatten_input:
            obs = obs.flatten(1)
        return self.model(obs)


class Net(nn.Module):
    def __init__(self, state_shape, action_shape=0, hidden_sizes=(), norm_layer=None,
                 activation=nn.ReLU, device="cpu", softmax=False, concat=False,
                 num_atoms=1, dueling_param=None, linear_layer=nn.Linear):
        super().__init__()
        self.device = device
        input_dim = int(np.prod(state_shape))
        action_dim = int(np.prod(action_shape)) * num_atoms
        if concat:
            input_dim += action_dim
        output_dim = action_dim if not concat else 0
        self.model = MLP(input_dim, output_dim, hidden_sizes, norm_layer, activation,
                         device, linear_layer)
        self.output_dim = self.model.output_dim

    def forward(self, obs, state=None, info={}):
        return self.model(obs), state


class ActorProb(nn.Module):
    def __init__(self, preprocess_net, action_shape, hidden_sizes=(), max_action=1.0,
                 device="cpu", unbounded=False, conditioned_sigma=False,
                 preprocess_net_output_dim=None):
        super().__init__()
        self.preprocess = preprocess_net
        self.device = device
        self.output_dim = int(np.prod(action_shape))
        input_dim = getattr(preprocess_net, "output_dim", preprocess_net_output_dim)
        self.mu = MLP(input_dim, self.output_dim, hidden_sizes, device=self.device)
        self._c_sigma = conditioned_sigma
        if conditioned_sigma:
            self.sigma = MLP(input_dim, self.output_dim, hidden_sizes, device=self.device)
        else:
            self.sigma_param = nn.Parameter(torch.zeros(self.output_dim, 1))
        self._max = max_action
        self._unbounded = unbounded

    def forward(self, obs, state=None, info={}):
        logits, hidden = self.preprocess(obs, state)
        mu = self.mu(logits)
        if not self._unbounded:
            mu = self._max * torch.tanh(mu)
        if self._c_sigma:
            sigma = torch.clamp(self.sigma(logits), min=SIGMA_MIN, max=SIGMA_MAX).exp()
        else:
            shape = [1] * len(mu.shape)
            shape[1] = -1
            sigma = (self.sigma_param.view(shape) + torch.zeros_like(mu)).exp()
        return (mu, sigma), state


class Actor(nn.Module):
    def __init__(self, preprocess_net, action_shape, hidden_sizes=(), max_action=1.0,
                 device="cpu", preprocess_net_output_dim=None):
        super().__init__()
        self.device = device
        self.preprocess = preprocess_net
        self.output_dim = int(np.prod(action_shape))
        input_dim = getattr(preprocess_net, "output_dim", preprocess_net_output_dim)
        self.last = MLP(input_dim, self.output_dim, hidden_sizes, device=self.device)
        self._max = max_action

    def forward(self, obs, state=None, info={}):
        logits, hidden = self.preprocess(obs, state)
        return self._max * torch.tanh(self.last(logits)), hidden


class Critic(nn.Module):
    def __init__(self, preprocess_net, hidden_sizes=(), device="cpu",
                 preprocess_net_output_dim=None, linear_layer=nn.Linear,
                 flatten_input=True):
        super().__init__()
        self.device = device
        self.preprocess = preprocess_net
        self.output_dim = 1
        input_dim = getattr(preprocess_net, "output_dim", preprocess_net_output_dim)
        self.last = MLP(input_dim, 1, hidden_sizes, device=self.device,
                        linear_layer=linear_layer, flatten_input=flatten_input)

    def forward(self, obs, act=None, info={}):
        obs = torch.as_tensor(obs, device=self.device, dtype=torch.float32).flatten(1)
        if act is not None:
            act = torch.as_tensor(act, device=self.device, dtype=torch.float32).flatten(1)
            obs = torch.cat([obs, act], dim=1)
        logits, hidden = self.preprocess(obs)
        return self.last(logits)


# ----------------------------------------------------------------------------- buffers
class ReplayBuffer:
    """Single ring buffer with tianshou-0.5 index semantics (only what FSRL touches)."""

    _keys = ("obs", "act", "rew", "terminated", "truncated", "done", "obs_next")

    def __init__(self, size, **kwargs):
        self.maxsize = int(size)
        self._meta = None
        self.reset()

    def reset(self, keep_statistics=False):
        self.last_index = np.array([0])
        self._index = self._size = 0
        if not keep_statistics:          # tianshou 0.5: the running episode's reward / length / start survive otherwise
            self._ep_rew, self._ep_len, self._ep_idx = 0.0, 0, 0

    def __len__(self):
        return self._size

    def _alloc(self, row):
        self._meta = {}
        for k, v in row.items():
            v = np.asarray(v)
            self._meta[k] = np.zeros((self.maxsize, ) + v.shape, dtype=v.dtype)

    def add_row(self, row):
        """row: dict key->value for ONE transition; returns (ptr, ep_rew, ep_len, ep_idx)."""
        if self._meta is None:
            self._alloc(row)
        ptr = self._index
        for k, v in row.items():
            self._meta[k][ptr] = v
        self.last_index[0] = ptr
        self._size = min(self._size + 1, self.maxsize)
        self._index = (self._index + 1) % self.maxsize
        self._ep_rew += float(row["rew"])
        self._ep_len += 1
        if row["done"]:
            out = (ptr, self._ep_rew, self._ep_len, self._ep_idx)
            self._ep_rew, self._ep_len, self._ep_idx = 0.0, 0, self._index
            return out
        return (ptr, 0.0, 0, self._ep_idx)

    def unfinished_index(self):
        last = (self._index - 1) % self._size if self._size else 0
        return np.array([last] if self._size and not self._meta["done"][last] else [], int)

    def next(self, index):
        index = np.asarray(index)
        end_flag = self._meta["done"][index] | (index == self.last_index[0])
        return (index + (1 - end_flag)) % self._size

    def sample_indices(self, batch_size):
        if batch_size > 0:
            return np.random.choice(self._size, batch_size)
        if batch_size == 0:
            return np.concatenate([np.arange(self._index, self._size), np.arange(self._index)])
        return np.array([], int)


class VectorReplayBuffer:
    """n sub-buffers of ceil(total/n) rows, contiguous offsets (ReplayBufferManager)."""

    def __init__(self, total_size, buffer_num, **kwargs):
        size = int(np.ceil(total_size / buffer_num))
        self.buffer_num = buffer_num
        self.buffers = [ReplayBuffer(size) for _ in range(buffer_num)]
        self._offset = np.array([i * size for i in range(buffer_num)])
        self.maxsize = size * buffer_num
        self._extend_offset = np.array(list(self._offset) + [self.maxsize])
        self._lengths = np.zeros_like(self._offset)
        self.last_index = self._offset.copy()
        self._meta = None

    def __len__(self):
        return int(self._lengths.sum())

    def reset(self, keep_statistics=False):
        self.last_index = self._offset.copy()
        self._lengths = np.zeros_like(self._offset)
        for b in self.buffers:
            b.reset(keep_statistics)

    def _alloc(self, row):
        self._meta = {}
        for k, v in row.items():
            v = np.asarray(v)
            self._meta[k] = np.zeros((self.maxsize, ) + v.shape, dtype=v.dtype)

    def add(self, rows, buffer_ids):
        """rows: dict key-> array over len(buffer_ids) transitions."""
        ptrs, ep_rews, ep_lens, ep_idxs = [], [], [], []
        for j, bid in enumerate(buffer_ids):
            row = {k: v[j] for k, v in rows.items()}
            if self._meta is None:
                self._alloc(row)
            ptr, r, l, i = self.buffers[bid].add_row({"rew": row["rew"], "done": row["done"]})
            gptr = ptr + self._offset[bid]
            for k, v in row.items():
                self._meta[k][gptr] = v
            self.last_index[bid] = gptr
            self._lengths[bid] = len(self.buffers[bid])
            ptrs.append(gptr); ep_rews.append(r); ep_lens.append(l)
            ep_idxs.append(i + self._offset[bid])
        return np.array(ptrs), np.array(ep_rews), np.array(ep_lens), np.array(ep_idxs)

    # attribute access to stored columns, like tianshou (buffer.rew, buffer.done, ...)
    def __getattr__(self, k):
        meta = self.__dict__.get("_meta")
        if meta is not None and k in meta:
            return meta[k]
        if meta is not None and k == "info":
            return Batch({kk[5:]: v for kk, v in meta.items() if kk.startswith("info.")})
        raise AttributeError(k)

    def __getitem__(self, index):
        index = np.asarray(index)
        b = Batch({k: v[index] for k, v in self._meta.items() if not k.startswith("info.")})
        b.info = Batch({k[5:]: v[index] for k, v in self._meta.items() if k.startswith("info.")})
        return b

    def unfinished_index(self):
        return np.concatenate([
            b.unfinished_index() + o for o, b in zip(self._offset, self.buffers)
        ]).astype(int)

    def next(self, index):
        index = np.asarray(index) % self.maxsize
        out = np.zeros_like(index)
        for o, nxt, b in zip(self._offset, self._extend_offset[1:], self.buffers):
            mask = (o <= index) & (index < nxt)
            if mask.any():
                out[mask] = b.next(index[mask] - o) + o
        return out

    def sample_indices(self, batch_size):
        if batch_size < 0:
            return np.array([], int)
        if batch_size == 0:
            return np.concatenate([
                b.sample_indices(0) + o for o, b in zip(self._offset, self.buffers)
            ])
        sample_num = np.random.choice(self.buffer_num, batch_size,
                                      p=self._lengths / self._lengths.sum())
        sample_num = np.bincount(sample_num, minlength=self.buffer_num)
        return np.concatenate([
            b.sample_indices(int(bsz)) + o
            for o, b, bsz in zip(self._offset, self.buffers, sample_num)
        ])

    def sample(self, batch_size):
        idx = self.sample_indices(batch_size)
        return self[idx], idx


# ----------------------------------------------------------------------------- numba
def njit(fn=None, **kw):
    """numba.njit stand-in that reproduces numba's scalar typing (python float -> float64,
    python int -> int64), so `f32_array * gamma` is a float64 multiply as under numba."""

    def deco(f):
        def wrapped(*args):
            conv = []
            for a in args:
                if isinstance(a, bool):
                    conv.append(a)
                elif isinstance(a, float):
                    conv.append(np.float64(a))
                elif isinstance(a, int):
                    conv.append(np.int64(a))
                else:
                    conv.append(a)
            return f(*conv)

        wrapped.__wrapped__ = f
        return wrapped

    return deco(fn) if callable(fn) else deco


# ----------------------------------------------------------------------------- install
def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Box:
    def __init__(self, low, high, shape=None, dtype=np.float32):
        shape = shape if shape is not None else np.shape(low)
        self.low = np.broadcast_to(np.asarray(low, dtype), shape).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype), shape).copy()
        self.shape = tuple(shape)
        self.dtype = dtype

    def sample(self):
        return np.random.uniform(self.low, self.high).astype(self.dtype)


def install(reference_root="/root/reference"):
    """Register the stand-in modules and put the reference on sys.path."""
    if "tianshou" in sys.modules and getattr(sys.modules["tianshou"], "_is_shim", False):
        return
    dummy = type("_Dummy", (), {})
    spaces = _mod("gymnasium.spaces", Box=_Box, Discrete=type("Discrete", (), {}),
                  MultiDiscrete=type("MultiDiscrete", (), {}),
                  MultiBinary=type("MultiBinary", (), {}), Space=object)
    _mod("gymnasium", spaces=spaces, Space=object, Env=object)
    _mod("numba", njit=njit)
    ts = _mod("tianshou", _is_shim=True)
    ts.data = _mod("tianshou.data", Batch=Batch, ReplayBuffer=ReplayBuffer,
                   ReplayBufferManager=VectorReplayBuffer,
                   VectorReplayBuffer=VectorReplayBuffer,
                   CachedReplayBuffer=dummy, PrioritizedReplayBuffer=dummy,
                   to_numpy=to_numpy, to_torch_as=to_torch_as)
    _mod("tianshou.data.utils")
    _mod("tianshou.data.utils.converter", to_hdf5=lambda *a, **k: None)

    class MultipleLRSchedulers:
        def __init__(self, *s):
            self.schedulers = s

        def step(self):
            for s in self.schedulers:
                s.step()

    class BaseNoise:
        def reset(self):
            pass

    class GaussianNoise(BaseNoise):
        def __init__(self, mu=0.0, sigma=1.0):
            self._mu, self._sigma = mu, sigma

        def __call__(self, size):
            return np.random.normal(self._mu, self._sigma, size)

    class DummyTqdm:
        def __init__(self, total=0, **kw):
            self.total, self.n = total, 0

        def set_postfix(self, **kw):
            pass

        def update(self, n=1):
            self.n += n

        def __enter__(self):
            return self

        def __exit__(self, *a):
            pass

    ts.utils = _mod("tianshou.utils", RunningMeanStd=RunningMeanStd,
                    MultipleLRSchedulers=MultipleLRSchedulers, DummyTqdm=DummyTqdm,
                    MovAvg=dummy, deprecation=lambda m: None,
                    tqdm_config={"dynamic_ncols": True, "ascii": True})
    _mod("tianshou.exploration", BaseNoise=BaseNoise, GaussianNoise=GaussianNoise)
    _mod("tianshou.env", BaseVectorEnv=dummy, DummyVectorEnv=dummy, ShmemVectorEnv=dummy,
         SubprocVectorEnv=dummy)
    _mod("tianshou.utils.net")
    _mod("tianshou.utils.net.common", MLP=MLP, Net=Net)
    _mod("tianshou.utils.net.continuous", ActorProb=ActorProb, Actor=Actor, Critic=Critic,
         SIGMA_MIN=SIGMA_MIN, SIGMA_MAX=SIGMA_MAX)
    _mod("wandb")
    tb = _mod("tensorboard")
    tb.backend = _mod("tensorboard.backend")
    tb.backend.event_processing = _mod("tensorboard.backend.event_processing",
                                       event_accumulator=dummy)
    try:
        import torch.utils.tensorboard  # noqa
    except Exception:
        _mod("torch.utils.tensorboard", SummaryWriter=dummy)
    _mod("h5py")
    if reference_root not in sys.path:
        sys.path.insert(0, reference_root)
