"""CPU-side checks of the drop-in boundary: the library loads and exports every symbol that
include/fsrl_hip.h declares (no compute calls -- there is no GPU in the build container)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "fsrl_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fsrl_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from fsrl_amd import _lib
    lib = _lib.load()
    syms = _header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/fsrl_hip.h but not exported"
    # and the binding table covers the header exactly
    assert sorted(_lib.SIGNATURES) == syms


def test_config_struct_matches_header_defaults():
    from fsrl_amd import _lib
    lib = _lib.load()
    c = _lib.Config()
    lib.fsrl_config_default(ctypes.byref(c))
    # reference defaults: fsrl/agent/ppo_lag_agent.py:82-116
    assert (c.obs_dim, c.act_dim, c.hidden, c.n_critics) == (8, 2, 128, 2)
    assert c.gamma == 0.99 and c.gae_lambda == 0.95
    assert abs(c.eps_clip - 0.2) < 1e-7 and abs(c.vf_coef - 0.25) < 1e-7
    assert abs(c.lr - 5e-4) < 1e-9 and abs(c.target_kl - 0.02) < 1e-8
    assert c.norm_adv == 1 and c.use_lagrangian == 1 and c.max_grad_norm == 0.0


def test_no_cpu_fallback_without_gpu():
    """The product path must fail loudly when no GPU is present (never route to the oracle)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from fsrl_amd._lib import FsrlHipError
    from fsrl_amd.engine import Engine, EngineConfig
    with pytest.raises(FsrlHipError):
        Engine(EngineConfig())


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "fsrl_amd")):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
