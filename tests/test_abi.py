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


def test_env_library_exports_every_declared_symbol():
    """include/fsrl_env.h vs libfsrl_env.so (the worker-process env's futex handshake), and one round trip through it on
    a private word pair: post -> wait_go sees the generation, done -> wait_done sees zero."""
    src = open(os.path.join(ROOT, "include", "fsrl_env.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    syms = sorted(set(re.findall(r"\b(fsrl_env_[a-z0-9_]+)\s*\(", src)))
    assert len(syms) == 6
    from fsrl_amd.env import shmem
    lib = shmem._load_lib()
    for s_ in syms:
        assert hasattr(lib, s_), f"{s_} declared in include/fsrl_env.h but not exported"
    words = (ctypes.c_uint32 * 64)()
    gen, pend = ctypes.addressof(words), ctypes.addressof(words) + 128
    lib.fsrl_env_post(gen, pend, 2, 7)
    assert words[0] == 7 and words[32] == 2
    assert lib.fsrl_env_wait_go(gen, 6, 10, 50) == 7                 # moved: returns at once
    assert lib.fsrl_env_wait_go(gen, 7, 10, 20) == 7                 # not moved: times out with the old value
    assert lib.fsrl_env_wait_done(pend, 10, 30) == -1                # two workers still pending
    lib.fsrl_env_done(pend); lib.fsrl_env_done(pend)
    assert lib.fsrl_env_wait_done(pend, 10, 30) == 0


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


def test_ctypes_structs_have_the_layout_the_c_compiler_gives_the_header(tmp_path):
    """Every configuration struct of include/fsrl_hip.h against its ctypes mirror in fsrl_amd/_lib.py: size and the offset of
    every field, as gcc lays the header out (a field added on one side only shifts everything behind it silently)."""
    import ctypes as C
    import re
    import subprocess
    from fsrl_amd import _lib
    pairs = {"fsrl_config": _lib.Config, "fsrl_tr_config": _lib.TrConfig, "fsrl_sac_config": _lib.SacConfig,
             "fsrl_focops_config": _lib.FocopsConfig, "fsrl_cvpo_config": _lib.CvpoConfig, "fsrl_shm_env": _lib.ShmEnv}
    lines = ["#include <stddef.h>", "#include <stdio.h>", '#include "fsrl_hip.h"', "int main(void) {"]
    for cname, cls in pairs.items():
        lines.append(f'  printf("{cname} size %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{cname} {fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = str(tmp_path / "layout")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run(["gcc", "-std=c99", "-I" + os.path.join(root, "include"), str(src), "-o", exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr            # a ctypes field the header does not have fails to compile here
    got = {}
    for l in subprocess.run([exe], capture_output=True, text=True).stdout.splitlines():
        cname, fname, val = l.split()
        got[(cname, fname)] = int(val)
    for cname, cls in pairs.items():
        assert got[(cname, "size")] == C.sizeof(cls), (cname, got[(cname, "size")], C.sizeof(cls))
        for fname, _ in cls._fields_:
            assert got[(cname, fname)] == getattr(cls, fname).offset, (cname, fname)
    # and the other direction: every field the header declares is mirrored (count of `;`-terminated members per struct)
    hdr = open(os.path.join(root, "include", "fsrl_hip.h")).read()
    for cname, cls in pairs.items():
        body = re.search(r"typedef struct " + cname + r" \{(.*?)\} " + cname + ";", hdr, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        n_members = sum(len(decl.split(",")) for decl in body.split(";") if decl.strip())
        assert n_members == len(cls._fields_), (cname, n_members, len(cls._fields_))
