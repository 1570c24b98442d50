"""CPU oracle for the FSRL policy-update hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under `fsrl_amd/` imports this package.  It is used by
`tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` as the
checker / the timed CPU port -- never as the thing shipped.

Pinning status (see DESIGN.md "Oracle"):
  * pinned against golden vectors generated from the UNMODIFIED reference code
    (`tests/golden/*.npz`, generator `tests/golden/gen_golden.py`) for: gae_return,
    nstep_return, LagrangianOptimizer, PPOLagrangian.update, CPO, TRPOLagrangian,
    SACLagrangian, DDPGLagrangian, FOCOPS and CVPO updates (generators tests/golden/gen_golden_*.py);
  * the tianshou~=0.5.0 pieces (Batch.split order, VectorReplayBuffer sample(0) order,
    MLP/ActorProb/Critic topology) are restated from that release's published behaviour;
    tianshou's source is not available in the build image => "parity unpinned" vs tianshou
    itself, pinned only against FSRL's call sites.
"""
from . import layout, pid, scans  # noqa: F401
