"""DeviceReplay (marl_amd/replay.py) and the device rollout loop (marl_amd/rollout.py) on the GPU:
the gather kernel against torch indexing, stepping-into-the-ring against plain stepping, and the
train_assembly.py-shaped loop end to end."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_gather_matches_torch_indexing(torch_cuda):
    torch = torch_cuda
    from marl_amd.batched import BatchedAssemblyEnv
    from marl_amd.replay import DeviceReplay
    env = BatchedAssemblyEnv(5, 30, seed=1)
    rp = DeviceReplay(env, episode_length=4, slots=3)
    g = torch.Generator(device="cuda").manual_seed(0)
    for t in (rp.obs_ring, rp.act_ring, rp.prior_ring, rp.rew_ring):
        t.copy_(torch.rand(t.shape, device="cuda", generator=g))
    rp.filled_slots = 3
    idx = rp.draw_indices(777)
    assert int(idx.min()) >= 0 and int(idx.max()) < len(rp)
    assert torch.equal(idx, DeviceReplayDraw(env, rp, 777)), "index draws are a pure function of (seed, counter)"
    obs, act, rew, nxt, done, pri = rp.gather(idx)
    E, n, T = env.E, env.n_a, rp.T
    i = idx % n; q = idx // n; e = q % E; st = q // E; s = st // T; t = st % T
    assert torch.equal(obs, rp.obs_ring[s, t, e, :, i])
    assert torch.equal(nxt, rp.obs_ring[s, t + 1, e, :, i])
    assert torch.equal(act, rp.act_ring[s, t, e, :, i])
    assert torch.equal(pri, rp.prior_ring[s, t, e, :, i])
    assert torch.equal(rew, rp.rew_ring[s, t, e, 0, i].unsqueeze(1))
    assert done.shape == (777, 1) and not done.any()
    # every transition id is reachable: the last id is the last agent of the last env, step and slot
    last = torch.tensor([len(rp) - 1], dtype=torch.int64, device="cuda")
    o = rp.gather(last)[0]
    assert torch.equal(o[0], rp.obs_ring[2, T - 1, E - 1, :, n - 1])


def DeviceReplayDraw(env, rp, N):
    """The same draw again: rewind the draw counter."""
    rp.draws -= 1
    return rp.draw_indices(N)


def test_stepping_into_the_ring_equals_plain_stepping(torch_cuda):
    torch = torch_cuda
    from marl_amd.batched import BatchedAssemblyEnv
    from marl_amd.replay import DeviceReplay
    E, n, T = 9, 30, 5
    a_env = BatchedAssemblyEnv(E, n, seed=4)
    b_env = BatchedAssemblyEnv(E, n, seed=4)
    rp = DeviceReplay(b_env, episode_length=T, slots=2)
    for ep in range(3):                                          # the third episode reuses slot 0
        rp.begin_episode()
        o_a = a_env.reset().clone(); o_b = b_env.reset()
        assert torch.equal(o_a, o_b) and o_b.data_ptr() == rp.obs_ring[rp.slot, 0].data_ptr()
        for t in range(T):
            act = rp.action_slot()
            a_env.random_actions(out=act, step=100 * ep + t)
            ra = a_env.step(act); rb = b_env.step(act)
            for x, y in zip((ra[0], ra[1], ra[4]), (rb[0], rb[1], rb[4])):
                assert torch.equal(x, y)
            assert rb[0].data_ptr() == rp.obs_ring[rp.slot, t + 1].data_ptr()
            rp.advance()
        assert torch.equal(a_env.episode_stats(), b_env.episode_stats())
    assert rp.filled_slots == 2 and len(rp) == 2 * T * E * n
    # a sampled transition's next_obs is the stored observation of the following step of the same agent
    obs, act, rew, nxt, done, pri = rp.sample(256)
    assert torch.isfinite(obs).all() and torch.isfinite(nxt).all() and set(np.unique(rew.cpu().numpy())) <= {0.0, 1.0}
    assert float(act.abs().max()) <= 1.0 and float(pri.abs().max()) <= 1.0 + 1e-6


def test_rollout_and_update_loop_runs(torch_cuda):
    torch = torch_cuda
    from marl_amd.batched import BatchedAssemblyEnv
    from marl_amd.replay import DeviceReplay
    from marl_amd.rollout import DDPGLearner, run_episode, train_phase
    from marl_amd.sharding import summarize
    torch.manual_seed(0)
    env = BatchedAssemblyEnv(16, 30, seed=2)
    rp = DeviceReplay(env, episode_length=6, slots=2)
    learner = DDPGLearner(env.obs_dim_agent, hidden_dim=32)
    before = [p.clone() for p in learner.policy.parameters()]
    for ep in range(2):
        stats = summarize(run_episode(env, learner, rp, 6), env.n_a)
        assert 0.0 <= stats["coverage"] <= 1.0 and np.isfinite(stats["reward_mean"])
        losses = train_phase(env, learner, rp, batch_size=512, num_updates=3)
        assert len(losses) == 3 and all(np.isfinite(v) for pair in losses for v in pair)
    assert any(not torch.equal(a, b) for a, b in zip(before, learner.policy.parameters()))
    assert learner.niter == 6


def test_expert_collection_fills_the_replay(torch_cuda):
    """collect_expert_episode: the rule expert drives the environments and its transitions are what the replay
    returns; over an episode the expert brings agents into their shapes (reward and coverage grow)."""
    torch = torch_cuda
    from marl_amd.batched import BatchedAssemblyEnv
    from marl_amd.replay import DeviceReplay
    from marl_amd.rollout import collect_expert_episode
    from marl_amd.sharding import summarize
    E, n_a, T = 12, 30, 60
    env = BatchedAssemblyEnv(E, n_a, seed=8, keep_indices=True)
    rp = DeviceReplay(env, episode_length=T, slots=1)
    stats = summarize(collect_expert_episode(env, rp, T), n_a)
    assert len(rp) == T * E * n_a
    assert stats["inside_fraction"] > 0.3 and stats["coverage"] > 0.05
    obs, acs, rews, nxt, dones, prior = rp.sample(4096)
    assert acs.abs().max().item() <= 1.0 and (acs.abs() == 1).any() and (acs.abs() < 1).any()
    assert not dones.any()
    # the expert's last action is reproducible from the state before the last step: replay one short episode by hand
    env2 = BatchedAssemblyEnv(E, n_a, seed=8, keep_indices=True)
    env2.reset(0)
    for _ in range(5):
        a = env2.rule_actions()
        env2.step(a)
    env3 = BatchedAssemblyEnv(E, n_a, seed=8, keep_indices=True)
    rp3 = DeviceReplay(env3, episode_length=5, slots=1)
    collect_expert_episode(env3, rp3, 5)
    assert torch.equal(env3.p, env2.p) and torch.equal(env3.obs, env2.obs)


def test_graphed_train_phase(torch_cuda):
    """GraphedTrainPhase: the captured update draws new transitions on every replay (device-side counter), moves the
    weights, and counts its updates like the eager loop; with the same seed it equals the eager loop's first update."""
    torch = torch_cuda
    from marl_amd.batched import BatchedAssemblyEnv
    from marl_amd.replay import DeviceReplay
    from marl_amd.rollout import DDPGLearner, GraphedTrainPhase, run_episode
    env = BatchedAssemblyEnv(16, 30, seed=2)
    rp = DeviceReplay(env, episode_length=6, slots=2)
    torch.manual_seed(0)
    learner = DDPGLearner(env.obs_dim_agent, hidden_dim=32, capturable=True)
    run_episode(env, learner, rp, 6)
    g = GraphedTrainPhase(env, learner, rp, batch_size=256)
    before = [p.detach().clone() for p in learner.critic.parameters()]
    seen = []
    losses = g.run(num_updates=3)                                   # the three warm-up updates of the capture
    seen.append(g.idx.clone())
    for _ in range(2):
        losses = torch.cat([losses, g.run(num_updates=4)])
        seen.append(g.idx.clone())
    assert losses.shape == (11, 2) and torch.isfinite(losses).all()
    assert learner.niter == 11 and int(g.counter.item()) == 11      # 3 warm-up updates + 8 replays (capturing runs nothing)
    assert not torch.equal(seen[0], seen[1]) and not torch.equal(seen[1], seen[2])
    assert int(seen[2].min()) >= 0 and int(seen[2].max()) < len(rp)
    assert any(not torch.equal(a, b) for a, b in zip(before, learner.critic.parameters()))
    run_episode(env, learner, rp, 6)                                # the replay grows: the graph is captured again
    n0 = learner.niter
    g.run(num_updates=5)
    assert g.captured_key == tuple(rp.complete) and learner.niter == n0 + 5


@pytest.mark.parametrize("E,n_a,obs_dim", [(5, 30, 192), (3, 7, 188), (2, 100, 192), (2, 1024, 192), (1, 33, 12)])
def test_agent_rows_equal_the_transposed_frames(torch_cuda, E, n_a, obs_dim):
    """asm_obs_rows_f32: rows[e][i][r] == obs[e][r][i], bit for bit (maddpg.py:84, the frame transposed for the policy)."""
    torch = torch_cuda
    from marl_amd import _lib
    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(E * 1000 + n_a)
    obs = torch.randn(E, obs_dim, n_a, device="cuda", generator=g)
    rows = torch.full((E, n_a, obs_dim), float("nan"), device="cuda")
    _lib.check(lib.asm_obs_rows_f32(E, n_a, obs_dim, obs.data_ptr(), rows.data_ptr(), torch.cuda.current_stream().cuda_stream), "rows")
    assert torch.equal(rows, obs.transpose(1, 2).contiguous())


def test_float32_act_uses_the_agent_rows(torch_cuda):
    """DDPGLearner.act in float32 (the reference's precision, networks.py:22-44) equals the network applied to the
    transposed frames, whichever way the rows were produced."""
    torch = torch_cuda
    from marl_amd.rollout import DDPGLearner
    torch.manual_seed(3)
    learner = DDPGLearner(192, hidden_dim=64)
    obs = torch.randn(6, 192, 30, device="cuda")
    out = torch.empty(6, 2, 30, device="cuda")
    learner.act(obs, out)
    with torch.no_grad():
        want = learner.policy(obs.transpose(1, 2).reshape(180, 192)).view(6, 30, 2).transpose(1, 2)
    assert torch.equal(out, want)


def test_draw_without_replacement_gives_distinct_ids(torch_cuda):
    """buffer_agent.py:149-154 draws its batch with np.random.choice(..., replace=False): the optional mode of the
    device draw returns N distinct ids of complete slots, down to N == every stored transition."""
    torch = torch_cuda
    from marl_amd.batched import BatchedAssemblyEnv
    from marl_amd.replay import DeviceReplay
    env = BatchedAssemblyEnv(3, 30)
    rp = DeviceReplay(env, episode_length=20, slots=3)
    rp.complete = [True, False, True]; rp._refresh_slot_table()       # 3600 stored rows, in slots 0 and 2
    per_slot = 20 * 3 * 30
    for N in (2048, 3600):                                            # a plain draw of 2048 out of 3600 repeats ~500 ids
        idx = rp.draw_indices(N, without_replacement=True)
        assert idx.numel() == N and torch.unique(idx).numel() == N
        assert bool(((idx // per_slot == 0) | (idx // per_slot == 2)).all())
    assert torch.unique(rp.draw_indices(2048)).numel() < 2048         # the default draws with replacement
    with pytest.raises(ValueError):
        rp.draw_indices(3601, without_replacement=True)
