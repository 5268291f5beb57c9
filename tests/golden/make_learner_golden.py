"""Records tests/golden/learner.npz from the reference's MADDPG class
(marl_llm/algorithm/algorithms/maddpg.py) run here in the container on the CPU:

  * one shared DDPG agent as init_from_env builds it for the assembly environment (observation 192, action 2;
    hidden width 32 to keep the file small), initial policy / critic weights stored;
  * one batch of 512 transitions (seeded; rewards in {0, 1}, dones 0, a third of the prior rows inside the
    |a| < 1e-2 mask of maddpg.py:162);
  * the weights of policy, critic and both targets after 3 x (update(..., acs_prior, alpha) + update_all_targets()),
    maddpg.py:99-199, with the reference's learning rates, gamma and tau (cfg/assembly_cfg.py:193-199).

    python tests/golden/make_learner_golden.py
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

sys.path.insert(0, "/root/reference/marl_llm")
from algorithm.algorithms.maddpg import MADDPG  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
OBS, ACT, HID, N, UPDATES, ALPHA = 192, 2, 32, 512, 3, 0.5

torch.manual_seed(7)
env = SimpleNamespace(observation_space=SimpleNamespace(shape=(OBS, 30)), action_space=SimpleNamespace(shape=(ACT, 30)),
                      agent_types=["agent"])
m = MADDPG.init_from_env(env, agent_alg="MADDPG", adversary_alg="MADDPG", tau=0.01, lr_actor=1e-4, lr_critic=1e-3,
                         hidden_dim=HID, device="cpu", epsilon=0.1, noise=0.9, name="assembly")
a = m.agents[0]
out = {"dims": np.array([OBS, ACT, HID, N, UPDATES]), "alpha": np.array(ALPHA), "gamma": np.array(m.gamma),
       "tau": np.array(m.tau), "lr_actor": np.array(1e-4), "lr_critic": np.array(1e-3)}
for net in ("policy", "critic"):
    for k, v in getattr(a, net).state_dict().items():
        out[f"init_{net}_{k}"] = v.numpy().copy()

rng = np.random.default_rng(3)
obs = rng.normal(0, 0.5, (N, OBS)).astype(np.float32)
nxt = (obs + rng.normal(0, 0.05, (N, OBS))).astype(np.float32)
acs = rng.uniform(-1, 1, (N, ACT)).astype(np.float32)
rews = rng.integers(0, 2, (N, 1)).astype(np.float32)
dones = np.zeros((N, 1), np.float32)
prior = rng.uniform(-1, 1, (N, ACT)).astype(np.float32)
prior[::3] = rng.uniform(-5e-3, 5e-3, (len(prior[::3]), ACT)).astype(np.float32)
out.update(obs=obs, next_obs=nxt, acs=acs, rews=rews, dones=dones, acs_prior=prior)

m.prep_training(device="cpu")
T = torch.from_numpy
for _ in range(UPDATES):
    m.update(T(obs), T(acs), T(rews), T(nxt), T(dones), 0, acs_prior=T(prior), alpha=ALPHA)
    m.update_all_targets()
for net in ("policy", "critic", "target_policy", "target_critic"):
    for k, v in getattr(a, net).state_dict().items():
        out[f"final_{net}_{k}"] = v.numpy().copy()
np.savez_compressed(os.path.join(HERE, "learner.npz"), **out)
print("wrote learner.npz", {k: v.shape for k, v in out.items() if k.startswith("final_policy")})
