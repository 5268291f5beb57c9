"""marl_amd.replay.ReplayBufferAgent against outputs of the reference's buffer class
(tests/golden/replay.npz, recorded by tests/golden/make_replay_golden.py): same pushes, same
numpy seed -> the same arrays and the same sampled batch, bit for bit."""
import os

import numpy as np
import torch

from marl_amd.replay import ReplayBufferAgent

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "replay.npz"))


def pushes(buf, rng, n_agents, state_dim, act_dim, count, index, with_prior):
    for _ in range(count):
        obs = rng.normal(size=(state_dim, n_agents)); act = rng.uniform(-1, 1, (act_dim, n_agents))
        rew = rng.integers(0, 2, (1, n_agents)).astype(np.float64); nxt = rng.normal(size=(state_dim, n_agents))
        done = np.zeros((1, n_agents), dtype=bool); pri = rng.uniform(-1, 1, (act_dim, n_agents))
        buf.push(obs, act, rew, nxt, done, index, pri if with_prior else None)


def test_push_with_tail_overwrite_matches_reference():
    b = ReplayBufferAgent(7, 3, slice(0, 3), 5, 2)
    pushes(b, np.random.default_rng(1), 3, 5, 2, 12, slice(0, 2), True)
    assert (b.curr_i, b.filled_i) == (int(G["small_curr"]), int(G["small_filled"]))
    for name, t in (("obs", b.obs_buffs), ("ac", b.ac_buffs), ("prior", b.ac_prior_buffs), ("rew", b.rew_buffs),
                    ("next", b.next_obs_buffs), ("done", b.done_buffs)):
        assert np.array_equal(t.numpy(), G["small_" + name]), name


def test_sample_matches_reference_under_the_same_numpy_seed():
    b = ReplayBufferAgent(100100, 3, slice(0, 3), 5, 2)
    rng = np.random.default_rng(3)
    n = b.total_length
    b.obs_buffs[:] = torch.from_numpy(rng.normal(size=(n, 5))); b.ac_buffs[:] = torch.from_numpy(rng.uniform(-1, 1, (n, 2)))
    b.ac_prior_buffs[:] = torch.from_numpy(rng.uniform(-1, 1, (n, 2)))
    b.rew_buffs[:] = torch.from_numpy(rng.integers(0, 2, (n, 1)).astype(np.float64))
    b.next_obs_buffs[:] = torch.from_numpy(rng.normal(size=(n, 5)))
    pushes(b, np.random.default_rng(2), 3, 5, 2, 40, slice(0, 3), True)
    assert (b.curr_i, b.filled_i) == (int(G["big_curr"]), int(G["big_filled"]))
    np.random.seed(5)
    s = b.sample(64, to_gpu=False, is_prior=True)
    assert s[6] is None and all(t.dtype == torch.float32 for t in s[:6])
    for name, t in zip(("obs", "ac", "rew", "next", "done", "prior"), s):
        assert np.array_equal(t.numpy(), G["s_" + name]), name
    assert np.abs(G["s_obs"]).sum() > 0
