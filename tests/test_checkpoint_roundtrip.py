"""f3: checkpoints written by the HIP path are the reference's wire format.

tests/golden/hip_ckpt_{ppo,sac}.pt were TRAINED ON AN MI355X through the engine and saved with policy.state_dict()
(tools/make_hip_checkpoint.py); tests/golden/hip_ckpt_ref_forward.npz holds what the UNMODIFIED reference policies return
after load_state_dict(strict=True) of those files (tests/golden/gen_ckpt_roundtrip.py, build container).

  * always: the host mirror (fsrl_amd.utils.net modules, no engine) loaded from the checkpoint gives the reference's
    outputs (<= 2e-6) and the outputs the DEVICE actor produced when the checkpoint was written (<= 1e-6);
  * where /root/reference exists (the build container): the reference-side load + forward is re-run live and must equal
    the committed fixture bit for bit."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from helpers import GOLDEN, load_npz


def _load(prefix, module, sd):
    sub = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    module.load_state_dict(sub, strict=True)


def test_ppo_checkpoint_host_mirror_equals_reference_forward():
    from fsrl_amd.utils.net import ActorProb, Critic, Net
    ck = torch.load(os.path.join(GOLDEN, "hip_ckpt_ppo.pt"), weights_only=False)
    ref = load_npz("hip_ckpt_ref_forward.npz")
    Do, Da, h = ck["net"]["obs_dim"], ck["net"]["act_dim"], tuple(ck["net"]["hidden"])
    actor = ActorProb(Net((Do, ), hidden_sizes=h), (Da, ), max_action=1.0)
    critics = [Critic(Net((Do, ), hidden_sizes=h)) for _ in range(2)]
    _load("actor.", actor, ck["model"])
    for i, c in enumerate(critics):
        _load(f"critics.{i}.", c, ck["model"])
    with torch.no_grad():
        (mu, sigma), _ = actor(ck["probe_obs"])
        vals = np.stack([c(ck["probe_obs"]).flatten().numpy() for c in critics])
    np.testing.assert_allclose(mu.numpy(), ref["ppo_mu"], rtol=0, atol=2e-6)          # == the reference after its own load
    np.testing.assert_allclose(sigma.numpy(), ref["ppo_sigma"], rtol=2e-6, atol=0)
    np.testing.assert_allclose(vals, ref["ppo_values"], rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(ref["ppo_mu"], ck["device_mu"], rtol=0, atol=1e-6)      # == what the MI355X computed
    np.testing.assert_allclose(ref["ppo_sigma"], ck["device_sigma"], rtol=1e-6, atol=0)
    np.testing.assert_allclose(ref["ppo_values"], ck["host_values"], rtol=2e-6, atol=2e-6)
    # a trained policy; the PID state travels in `_extra_state` exactly as the reference writes it
    # (lagrangian_base.py:122-131).  The reference's own set_extra_state (:133-143) looks for a nested "_extra_state" key
    # inside the list torch hands it, so IT restores nothing -- from its own checkpoints either; the facade restores it.
    assert ck["gradient_steps"] > 0 and ck["model"]["_extra_state"][0]["lagrangian"] > 0
    assert set(ck["model"]["_extra_state"][0]) == {"pid", "error_old", "error_integral", "lagrangian"}
    assert ref["ppo_lagrangian"][0] == 0.0
    # the flat vector the engine held == the checkpoint, tensor by tensor in parameters() order
    flat = np.concatenate([ck["model"][k].numpy().reshape(-1) for k in ck["model"]
                           if k.startswith("actor.") or k.startswith("critics.")])
    assert np.array_equal(flat, ck["flat_params"])


def test_sac_checkpoint_host_mirror_equals_reference_forward():
    from fsrl_amd.utils.net import ActorProb, DoubleCritic, Net
    ck = torch.load(os.path.join(GOLDEN, "hip_ckpt_sac.pt"), weights_only=False)
    ref = load_npz("hip_ckpt_ref_forward.npz")
    Do, Da, h = ck["net"]["obs_dim"], ck["net"]["act_dim"], tuple(ck["net"]["hidden"])
    actor = ActorProb(Net((Do, ), hidden_sizes=h), (Da, ), conditioned_sigma=True, unbounded=True)
    mk = lambda: DoubleCritic(Net((Do, ), (Da, ), hidden_sizes=h, concat=True), Net((Do, ), (Da, ), hidden_sizes=h, concat=True))  # noqa: E731
    critics, old = [mk(), mk()], [mk(), mk()]
    _load("actor.", actor, ck["model"])
    for i in range(2):
        _load(f"critics.{i}.", critics[i], ck["model"])
        _load(f"critics_old.{i}.", old[i], ck["model"])
    with torch.no_grad():
        (mu, sigma), _ = actor(ck["probe_obs"])
        q = np.array([[x.flatten().numpy() for x in c(ck["probe_obs"], ck["probe_act"])] for c in critics])
        qo = np.array([[x.flatten().numpy() for x in c(ck["probe_obs"], ck["probe_act"])] for c in old])
    np.testing.assert_allclose(mu.numpy(), ref["sac_mu"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(sigma.numpy(), ref["sac_sigma"], rtol=2e-6, atol=0)
    np.testing.assert_allclose(q, ref["sac_q"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(qo, ref["sac_q_old"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(ref["sac_mu"], ck["device_mu"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(ref["sac_sigma"], ck["device_sigma"], rtol=1e-6, atol=0)
    # critics / targets in the file are the DEVICE's (pulled after an actor-only host forward: ADVICE r1, high)
    for key, pre in (("flat_critics", "critics."), ("flat_critics_old", "critics_old."), ("flat_actor", "actor.")):
        flat = np.concatenate([ck["model"][k].numpy().reshape(-1) for k in ck["model"] if k.startswith(pre)])
        assert np.array_equal(flat, ck[key]), key
    assert not np.array_equal(ck["flat_critics"], ck["flat_critics_old"]) and ck["gradient_steps"] > 0


@pytest.mark.skipif(not os.path.isdir("/root/reference/fsrl"), reason="the reference tree exists in the build container only")
def test_unmodified_reference_loads_the_hip_checkpoints_live(tmp_path):
    out = str(tmp_path / "live.npz")
    r = subprocess.run([sys.executable, os.path.join(GOLDEN, "gen_ckpt_roundtrip.py"), out], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.count("strict load ok <All keys matched successfully>") == 2
    live, ref = dict(np.load(out)), load_npz("hip_ckpt_ref_forward.npz")
    assert set(live) == set(ref)
    for k in ref:
        assert np.array_equal(live[k], ref[k]), k
