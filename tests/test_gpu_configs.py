"""The three BASELINE configurations beyond the 4096-environment benchmark, at full per-GPU size on one GPU:

* the large swarm, 256 environments x 1024 agents stepped as two stream shards (one 1024-thread workgroup on
  every CU with both streams busy), sampled environments against the oracle every step;

* one rank's shard of "65536 envs x 30 agents sharded over 8 GPUs": rank 7's 8192 environments (global ids
  57344..65535) stepped as two stream shards, with the episode-statistics vector that rank would all-reduce;
* the train_assembly.py-shaped loop (train_assembly.py:99-157) at 4096 environments x 200 steps with the replay
  in HBM and the reference's network width."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_config3_rank_shard_full_size(torch_cuda):
    torch = torch_cuda
    import oracle_lib as ol
    from marl_amd.batched import BatchedAssemblyEnv
    from marl_amd.sharding import StreamShards, shard_range
    from gpu_util import cfg_from_env, teacher_forced_step
    n_a, steps = 30, 20
    first, E = shard_range(7, 8, 8192)
    assert (first, E) == (57344, 8192)
    sh = StreamShards(E, n_a, parts=2, seed=226, env_id_offset=first, keep_indices=True)
    assert [e.env_id_offset for e in sh.envs] == [57344, 57344 + 4096]
    # a twin of 24 environments spread over both stream shards, checked against the oracle every step
    pick = np.concatenate([np.arange(0, 12), np.arange(4096 + 100, 4096 + 112)])
    twins = [BatchedAssemblyEnv(12, n_a, seed=226, env_id_offset=first + int(pick[0]), keep_indices=True),
             BatchedAssemblyEnv(12, n_a, seed=226, env_id_offset=first + int(pick[12]), keep_indices=True)]
    cfg = cfg_from_env(twins[0])
    sh.reset(); sh.synchronize()
    for tw in twins:
        tw.reset()
    want = np.zeros(8)
    for t in range(steps):
        acts = sh.on_each(lambda k, env: env.random_actions(step=t))
        sh.step(acts); sh.synchronize()
        for k, (tw, lo) in enumerate(zip(twins, (0, 100))):
            env = sh.envs[k]
            a = acts[k][lo:lo + 12]
            o, contacts = teacher_forced_step(tw, cfg, a.cpu().numpy(), f"shard {k} step {t}")   # HIP twin == oracle
            for name in ("obs", "reward", "a_prior", "p", "dp", "neighbor_index", "in_flags", "sensed_index"):
                assert torch.equal(getattr(env, name)[lo:lo + 12], getattr(tw, name)), (name, k, t)   # shard == twin, bit for bit
            p1 = tw.p.cpu().numpy().astype(np.float64); grid = tw.grid_center.cpu().numpy(); n_g = tw.n_g.cpu().numpy()
            for e in range(12):
                g = np.ascontiguousarray(grid[e][:, :n_g[e]], dtype=np.float64)
                m = ol.metrics(np.ascontiguousarray(p1[e]), g, tw.r_avoid)
                want += [o["rew"][e].sum(), (o["rew"][e] ** 2).sum(), m["coverage_rate"], m["distribution_uniformity"],
                         contacts[e], m["avg_distance_to_target"], 1.0, o["in_flags"][e].sum()]
        for env in sh.envs:                                  # size-independent properties over every environment
            assert torch.isfinite(env.obs).all() and float(env.p.abs().max()) <= 2.5
            assert float(env.dp.abs().max()) <= float(np.float32(0.8))
            assert set(torch.unique(env.reward).tolist()) <= {0.0, 1.0}
            s = env.sensed_index; valid = s >= 0
            assert bool((valid[..., 1:] <= valid[..., :-1]).all())
            assert bool(((s[..., 1:] > s[..., :-1]) | ~valid[..., 1:]).all())
    # the statistics vector of the sample against the oracle's sums, and of the whole shard
    got = sh.envs[0].stats[0:12].sum(0) + sh.envs[1].stats[100:112].sum(0)
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-5, atol=1e-9)
    st = sh.episode_stats().cpu().numpy()
    assert st[6] == steps * E and 0 <= st[0] <= steps * E * n_a and st[0] == st[1]
    assert st[0] == pytest.approx(float(sum(env.stats[:, 0].sum() for env in sh.envs)))
    assert 0.0 <= st[2] / st[6] <= 1.0 and st[7] <= steps * E * n_a


def test_config4_large_swarm_full_size(torch_cuda):
    torch = torch_cuda
    import oracle_lib as ol
    from marl_amd.batched import BatchedAssemblyEnv
    from marl_amd.sharding import StreamShards
    from gpu_util import cfg_from_env, teacher_forced_step
    E, n_a, steps, S = 256, 1024, 10, 2
    sh = StreamShards(E, n_a, parts=2, seed=226, keep_indices=True)
    assert [e.E for e in sh.envs] == [128, 128] and sh.envs[0].step_schedule() & 15 == 0       # the large-swarm kernel
    # twins of two environments per stream shard (a pair at each end of the batch), checked against the oracle every step
    lows = (0, 126)
    twins = [BatchedAssemblyEnv(S, n_a, seed=226, env_id_offset=k * 128 + lo, keep_indices=True) for k, lo in enumerate(lows)]
    assert all(tw.step_schedule() == sh.envs[0].step_schedule() for tw in twins)
    cfg = cfg_from_env(twins[0])
    assert sh.envs[1].r_avoid == twins[0].r_avoid and abs(twins[0].r_avoid - 0.05) < 1e-12     # assembly.py:122 at 1024 agents
    sh.reset(); sh.synchronize()
    for tw in twins:
        tw.reset()
    # the reset has placed the shapes, so their lattice description is in use: lattice path, cell lists, the large swarm's
    # own instantiation of the generic kernel -- for the shards and for the twins
    assert all(x.step_schedule() == 32 + 64 + 256 for x in list(sh.envs) + twins)
    want = np.zeros(8)
    for t in range(steps):
        acts = sh.on_each(lambda k, env: env.random_actions(step=t))
        sh.step(acts); sh.synchronize()
        for k, (tw, lo) in enumerate(zip(twins, lows)):
            env = sh.envs[k]
            o, contacts = teacher_forced_step(tw, cfg, acts[k][lo:lo + S].cpu().numpy(), f"shard {k} step {t}")   # HIP twin == oracle
            for name in ("obs", "reward", "a_prior", "p", "dp", "neighbor_index", "in_flags", "sensed_index", "occupied_index"):
                assert torch.equal(getattr(env, name)[lo:lo + S], getattr(tw, name)), (name, k, t)   # shard == twin, bit for bit
            p1 = tw.p.cpu().numpy().astype(np.float64); grid = tw.grid_center.cpu().numpy(); n_g = tw.n_g.cpu().numpy()
            for e in range(S):
                g = np.ascontiguousarray(grid[e][:, :n_g[e]], dtype=np.float64)
                m = ol.metrics(np.ascontiguousarray(p1[e]), g, tw.r_avoid)
                want += [o["rew"][e].sum(), (o["rew"][e] ** 2).sum(), m["coverage_rate"], m["distribution_uniformity"],
                         contacts[e], m["avg_distance_to_target"], 1.0, o["in_flags"][e].sum()]
        for env in sh.envs:                                  # size-independent properties over every environment
            assert torch.isfinite(env.obs).all() and float(env.p.abs().max()) <= 2.5
            assert float(env.dp.abs().max()) <= float(np.float32(0.8))
            assert set(torch.unique(env.reward).tolist()) <= {0.0, 1.0}
            nb = env.neighbor_index
            assert int(nb.max()) < n_a and int(nb.min()) >= -1
            assert not bool((nb == torch.arange(n_a, device=nb.device)[None, :, None]).any())    # never its own neighbour (CPP:677-681)
            s = env.sensed_index; valid = s >= 0
            assert bool((valid[..., 1:] <= valid[..., :-1]).all())
            assert bool(((s[..., 1:] > s[..., :-1]) | ~valid[..., 1:]).all())
            assert bool((env.in_flags.bool() | ~(env.occupied_index >= 0).any(-1)).all())        # occupied cells only inside the shape (CPP:155)
    got = sum(sh.envs[k].stats[lo:lo + S].sum(0) for k, lo in enumerate(lows))
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-5, atol=1e-9)
    st = sh.episode_stats().cpu().numpy()
    assert st[6] == steps * E and 0 <= st[0] <= steps * E * n_a and st[0] == st[1]
    assert 0.0 <= st[2] / st[6] <= 1.0 and st[7] <= steps * E * n_a and st[4] > 0              # 1024 agents: contacts happen


def test_config5_rollout_and_training_full_size(torch_cuda):
    torch = torch_cuda
    from marl_amd.batched import BatchedAssemblyEnv
    from marl_amd.replay import DeviceReplay
    from marl_amd.rollout import DDPGLearner, run_episode, train_phase
    from marl_amd.sharding import summarize
    E, n_a, T = 4096, 30, 200
    torch.manual_seed(0)
    env = BatchedAssemblyEnv(E, n_a, seed=226)
    rp = DeviceReplay(env, episode_length=T, slots=2)           # 2 x 19 GB of observations in HBM
    learner = DDPGLearner(env.obs_dim_agent, hidden_dim=256, act_dtype=torch.bfloat16)   # cfg/assembly_cfg.py:196
    gen = torch.Generator(device="cuda").manual_seed(5)
    twin = BatchedAssemblyEnv(E, n_a, seed=226)
    per_slot = T * E * n_a
    for ep in range(3):                                          # the third episode rewrites slot 0
        stats = run_episode(env, learner, rp, T, noise_scale=0.9, epsilon=0.1, generator=gen)
        assert rp.slot == ep % 2 and rp.t == T and len(rp) == min(ep + 1, 2) * per_slot
        s = summarize(stats, n_a)
        assert stats[6].item() == T * E and 0.0 <= s["coverage"] <= 1.0 and 0.0 <= s["reward_mean"] <= 1.0
        # plain stepping with the stored actions reproduces the stored frames and the statistics
        twin.reset(episode=ep)
        assert torch.equal(twin.obs, rp.obs_ring[rp.slot, 0])
        for t in range(T):
            o, r, _, _, pr = twin.step(rp.act_ring[rp.slot, t])
            if t % 40 == 0 or t == T - 1:
                assert torch.equal(o, rp.obs_ring[rp.slot, t + 1]) and torch.equal(r, rp.rew_ring[rp.slot, t])
                assert torch.equal(pr, rp.prior_ring[rp.slot, t])
        assert torch.equal(twin.episode_stats(), stats)
        before = [p.detach().clone() for p in learner.critic.parameters()]
        losses = train_phase(env, learner, rp, batch_size=2048)  # 30 updates of 2048 rows (train_assembly.py:144-157)
        assert len(losses) == 30 and all(np.isfinite(v) for pair in losses for v in pair)
        assert any(not torch.equal(a, b) for a, b in zip(before, learner.critic.parameters()))
    assert learner.niter == 90
    # while slot 1 is being rewritten only slot 0 is sampled
    rp.begin_episode(); env.reset()
    assert rp.slot == 1 and len(rp) == per_slot
    idx = rp.draw_indices(4096)
    assert int(idx.min()) >= 0 and int(idx.max()) < per_slot
    a = rp.action_slot(); a.zero_(); env.step(a); rp.advance()
    obs, acs, rews, nxt, dones, prior = rp.sample(2048)
    assert torch.isfinite(obs).all() and torch.isfinite(nxt).all() and float(acs.abs().max()) <= 1.0
