#!/usr/bin/env python
"""Produce HIP-trained checkpoints for the reference-side round trip (fsrl/utils/exp_util.py:60-84,
fsrl/agent/base_agent.py:75-76).  Runs on the GPU box:

    python tools/make_hip_checkpoint.py gpurun_out/ckpt

trains a small PPO-Lagrangian and a small SAC-Lagrangian agent on the synthetic env THROUGH THE HIP ENGINE, then writes
`hip_ckpt_ppo.pt` / `hip_ckpt_sac.pt`: {"model": policy.state_dict()} exactly as the agents' checkpoint_fn saves it
(base_agent.py:293-294), plus probe observations and what the DEVICE actor returns for them (fsrl_actor_forward /
fsrl_sac_actor_forward: evidence that the checkpoint holds the device's parameters) and the host mirror's critic outputs.
The files are copied to tests/golden/ and loaded into the unmodified reference by tests/golden/gen_ckpt_roundtrip.py
in the build container."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(outdir):
    from fsrl_amd.agent import PPOLagAgent, SACLagAgent
    from fsrl_amd.env import SyntheticSafetyVectorEnv
    os.makedirs(outdir, exist_ok=True)
    probe = np.random.default_rng(5).standard_normal((32, 8)).astype(np.float32)
    # ---- PPO-Lagrangian, obs 8 / act 2 / 64x64
    env = SyntheticSafetyVectorEnv(env_num=4, obs_dim=8, act_dim=2, episode_len=50, seed=3)
    agent = PPOLagAgent(env, None, cost_limit=10, device="cuda:0", seed=7, hidden_sizes=(64, 64), training_num=4)
    agent.learn(env, None, epoch=3, episode_per_collect=8, step_per_epoch=800, repeat_per_collect=2, batch_size=64,
                verbose=False, save_ckpt=False, show_progress=False)
    pol = agent.policy
    sd = pol.state_dict()                                   # pulls the device parameters into the host mirror
    mu, sigma = pol.engine.actor_forward(probe)
    with torch.no_grad():
        vals = [c(probe).flatten().numpy() for c in pol.critics]
    torch.save({"model": sd, "probe_obs": probe, "device_mu": mu, "device_sigma": sigma, "host_values": np.stack(vals),
                "flat_params": pol.engine.get_params(), "gradient_steps": pol.gradient_steps,
                "net": dict(obs_dim=8, act_dim=2, hidden=[64, 64])}, os.path.join(outdir, "hip_ckpt_ppo.pt"))
    print("ppo checkpoint:", pol.gradient_steps, "gradient steps, lagrangian", pol.lag_optims[0].get_lag())
    pol.engine.close()
    # ---- SAC-Lagrangian, obs 8 / act 2 / 64x64
    env = SyntheticSafetyVectorEnv(env_num=4, obs_dim=8, act_dim=2, episode_len=50, seed=4)
    agent = SACLagAgent(env, None, cost_limit=10, device="cuda:0", seed=7, hidden_sizes=(64, 64), training_num=4,
                        buffer_size=4000)
    agent.learn(env, None, epoch=2, episode_per_collect=4, step_per_epoch=400, update_per_step=0.25, batch_size=64,
                verbose=False, save_ckpt=False, show_progress=False)
    pol = agent.policy
    pol(__import__("fsrl_amd.data.batch", fromlist=["Batch"]).Batch(obs=probe, info={}))     # an actor-only pull first ...
    sd = pol.state_dict()                                   # ... state_dict must still fetch critics and targets
    mu, sigma = pol.engine.sac_actor_forward(probe)
    act = np.tanh(mu)
    with torch.no_grad():
        q = [[x.flatten().numpy() for x in c(probe, act)] for c in pol.critics]
        qo = [[x.flatten().numpy() for x in c(probe, act)] for c in pol.critics_old]
    torch.save({"model": sd, "probe_obs": probe, "probe_act": act, "device_mu": mu, "device_sigma": sigma,
                "host_q": np.array(q), "host_q_old": np.array(qo),
                "flat_actor": pol.engine.sac_get_params(0)[0], "flat_critics": pol.engine.sac_get_params(1)[0],
                "flat_critics_old": pol.engine.sac_get_params(2)[0], "gradient_steps": pol.gradient_steps,
                "net": dict(obs_dim=8, act_dim=2, hidden=[64, 64])}, os.path.join(outdir, "hip_ckpt_sac.pt"))
    print("sac checkpoint:", pol.gradient_steps, "gradient steps")
    pol.engine.close()


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "ckpt"))
