"""DDPGLearner (marl_amd/rollout.py) against the reference's MADDPG class: tests/golden/learner.npz holds initial
weights, one batch and the weights after 3 x (update + update_all_targets) recorded from
marl_llm/algorithm/algorithms/maddpg.py:99-199 on the CPU by tests/golden/make_learner_golden.py.  Runs on the CPU
(the learner is PyTorch code; the environment is not involved).

Tolerance: both sides run the same float32 torch operators on the same machine class; 1e-6 relative / 1e-7
absolute allows for a different reduction order inside a BLAS call and nothing more."""
import os

import numpy as np
import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RTOL, ATOL = 1e-6, 1e-7
NETS = ("policy", "critic", "target_policy", "target_critic")


def _load(z, learner):
    for net in ("policy", "critic"):
        sd = {k: torch.from_numpy(z[f"init_{net}_{k}"]) for k in getattr(learner, net).state_dict()}
        getattr(learner, net).load_state_dict(sd)
    learner.target_policy.load_state_dict(learner.policy.state_dict())
    learner.target_critic.load_state_dict(learner.critic.state_dict())


def test_three_updates_match_reference_maddpg():
    from marl_amd.rollout import DDPGLearner
    z = np.load(os.path.join(GOLDEN, "learner.npz"))
    obs_dim, act_dim, hid, n, updates = (int(v) for v in z["dims"])
    torch.manual_seed(0)
    torch.set_num_threads(1)
    L = DDPGLearner(obs_dim, act_dim, hidden_dim=hid, lr_actor=float(z["lr_actor"]), lr_critic=float(z["lr_critic"]),
                    gamma=float(z["gamma"]), tau=float(z["tau"]), device="cpu")
    _load(z, L)
    T = lambda k: torch.from_numpy(z[k])
    assert int((np.abs(z["acs_prior"]) < 1e-2).all(axis=1).sum()) > 100        # the mask of maddpg.py:162 is exercised
    for _ in range(updates):
        vf, pl = L.update(T("obs"), T("acs"), T("rews"), T("next_obs"), T("dones"), T("acs_prior"), float(z["alpha"]))
        assert np.isfinite(vf) and np.isfinite(pl)
        L.update_all_targets()
    assert L.niter == updates
    for net in NETS:
        for k, v in getattr(L, net).state_dict().items():
            want = z[f"final_{net}_{k}"]
            np.testing.assert_allclose(v.numpy(), want, rtol=RTOL, atol=ATOL, err_msg=f"{net}.{k}")
            assert not np.array_equal(want, z.get(f"init_{net.replace('target_', '')}_{k}")), "weights moved"


def test_update_without_valid_prior_rows_adds_nothing():
    """maddpg.py:168-169: a batch whose prior rows all lie inside the mask contributes no regularisation."""
    from marl_amd.rollout import DDPGLearner
    z = np.load(os.path.join(GOLDEN, "learner.npz"))
    obs_dim, act_dim, hid, n, _ = (int(v) for v in z["dims"])
    T = lambda k: torch.from_numpy(z[k])
    outs = []
    for prior in (torch.zeros(n, act_dim), None):
        L = DDPGLearner(obs_dim, act_dim, hidden_dim=hid, device="cpu")
        _load(z, L)
        L.update(T("obs"), T("acs"), T("rews"), T("next_obs"), T("dones"), prior, 0.5)
        outs.append([p.detach().clone() for p in L.policy.parameters()])
    for a, b in zip(*outs):
        assert torch.equal(a, b)
