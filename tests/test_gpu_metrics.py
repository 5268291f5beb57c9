"""GPU tests of the evaluation-side entry points of the batched environment: the formation metrics
(assembly_wrapper.py:48-187) as one device call, and mid-episode shape switching (eval_assembly.py:17-57)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _oracle_metrics(env):
    import oracle_lib as ol
    p = env.p.cpu().numpy().astype(np.float64); grid = env.grid_center.cpu().numpy(); n_g = env.n_g.cpu().numpy()
    want = np.zeros((env.E, 5))
    for e in range(env.E):
        g = np.ascontiguousarray(grid[e][:, :n_g[e]], dtype=np.float64)
        m = ol.metrics(np.ascontiguousarray(p[e]), g, env.r_avoid)
        dx = p[e][0][None, :] - p[e][0][:, None]; dy = p[e][1][None, :] - p[e][1][:, None]
        pairs = np.triu(np.sqrt(dx * dx + dy * dy) - 2 * 0.035 < 0, 1).sum()        # assembly.py:445-446
        want[e] = [m["coverage_rate"], m["distribution_uniformity"], m["voronoi_uniformity"], pairs,
                   m["avg_distance_to_target"]]
    return want


@pytest.mark.parametrize("n_a,E", [(30, 24), (7, 5), (100, 4), (1024, 2)])
def test_formation_metrics_match_oracle(torch_cuda, n_a, E):
    """Every metric of every environment equals the oracle's restatement of the wrapper bit for bit (same
    correctly rounded square roots, same left-to-right sums), in the random and the packed regime."""
    torch = torch_cuda
    from marl_amd.batched import BatchedAssemblyEnv
    env = BatchedAssemblyEnv(E, n_a, seed=3 + n_a)
    env.reset()
    rng = np.random.default_rng(n_a)
    grid = env.grid_center.cpu().numpy(); n_g = env.n_g.cpu().numpy(); p = env.p.cpu().numpy()
    for e in range(0, E, 2):                                    # every other environment packed into its shape
        cells = rng.choice(n_g[e], n_a, replace=n_a > n_g[e])
        p[e] = grid[e][:, cells] + rng.normal(0, 0.02, (2, n_a))
    env.p.copy_(torch.from_numpy(p)); env.observe()
    for t in range(3):
        got = env.formation_metrics().cpu().numpy()
        want = _oracle_metrics(env)
        np.testing.assert_array_equal(got, want, err_msg=f"n_a {n_a} step {t}")
        assert (got[::2, 0] > 0).all()                          # packed environments do cover cells
        env.step(env.random_actions())
    assert env.lib.asm_batch_metrics_f64(None, E, None, None, None, None, None) != 0


def test_metrics_agree_with_step_statistics(torch_cuda):
    """The step kernel accumulates coverage / uniformity / average distance per step (fp32-screened); the
    metrics call on the same state must give the same numbers to 1e-5."""
    from marl_amd.batched import BatchedAssemblyEnv
    env = BatchedAssemblyEnv(16, 30, seed=5)
    env.reset()
    for _ in range(4):
        before = env.stats.clone()
        pre = env.formation_metrics().cpu().numpy()             # contact pairs of the state the step starts from
        env.step(env.random_actions())
        d = (env.stats - before).cpu().numpy(); m = env.formation_metrics().cpu().numpy()
        np.testing.assert_allclose(d[:, 2], m[:, 0], rtol=1e-12)
        np.testing.assert_allclose(d[:, 3], m[:, 1], rtol=1e-5)
        np.testing.assert_allclose(d[:, 5], m[:, 4], rtol=1e-5)
        np.testing.assert_array_equal(d[:, 4], pre[:, 3])


def test_shape_switch_mid_episode(torch_cuda):
    """process_shape of eval_assembly.py: some environments get a new canonical shape between two steps; the
    following steps (prior included, which must search its target on the new grid) match the oracle."""
    torch = torch_cuda
    from marl_amd.batched import BatchedAssemblyEnv
    from gpu_util import cfg_from_env, check_observation, teacher_forced_step
    E, n_a = 8, 30
    env = BatchedAssemblyEnv(E, n_a, seed=17, keep_indices=True)
    cfg = cfg_from_env(env)
    env.reset()
    rng = np.random.default_rng(2)
    teacher_forced_step(env, cfg, rng.uniform(-1, 1, (E, 2, n_a)).astype(np.float32), "before")
    ids = torch.tensor([1, 4, 6])
    old_grid = env.grid_center.clone()
    env.set_shape(4, ids)
    k = 4
    assert (env.n_g[ids].cpu().numpy() == env.n_gs[k]).all() and (env.shape_id[ids].cpu().numpy() == k).all()
    np.testing.assert_array_equal(env.grid_center[1, :, :env.n_gs[k]].cpu().numpy(),
                                  env.grid_center_origins[k].T.astype(np.float32))
    keep = [0, 2, 3, 5, 7]
    assert torch.equal(env.grid_center[keep], old_grid[keep])
    for t in range(3):
        teacher_forced_step(env, cfg, rng.uniform(-1, 1, (E, 2, n_a)).astype(np.float32), f"after switch {t}")
    env.set_shape(0)                                            # all environments
    assert (env.n_g.cpu().numpy() == env.n_gs[0]).all()
    teacher_forced_step(env, cfg, rng.uniform(-1, 1, (E, 2, n_a)).astype(np.float32), "after second switch")
    with pytest.raises(IndexError):
        env.set_shape(99)


def test_eval_episode_with_shape_schedule(torch_cuda):
    """The evaluation loop (eval_assembly.py:115-190): metrics recorded before every step, shapes switched at the
    scheduled steps; the recorded rows equal a hand-driven replay of the same episode."""
    torch = torch_cuda
    from marl_amd.batched import BatchedAssemblyEnv
    from marl_amd.rollout import DDPGLearner, eval_episode
    E, n_a, T = 6, 30, 9
    torch.manual_seed(0)
    env = BatchedAssemblyEnv(E, n_a, seed=21)
    learner = DDPGLearner(env.observation_space.shape[0], hidden_dim=32)
    metrics, total = eval_episode(env, learner, T, {0: 4, 3: 5, 6: 6})
    assert metrics.shape == (T, E, 5) and total.shape == (E,)
    assert (env.shape_id.cpu().numpy() == 6).all()
    env2 = BatchedAssemblyEnv(E, n_a, seed=21)
    obs = env2.reset(0); a = torch.empty(E, 2, n_a, device="cuda"); tot = torch.zeros(E, dtype=torch.float64, device="cuda")
    for t in range(T):
        if t in (0, 3, 6):
            env2.set_shape({0: 4, 3: 5, 6: 6}[t])
        assert torch.equal(env2.formation_metrics(), metrics[t]) or \
            np.array_equal(env2.formation_metrics().cpu().numpy(), metrics[t].cpu().numpy(), equal_nan=True)
        obs, rew = env2.step(learner.act(obs, a))[:2]
        tot += rew.sum(dim=(1, 2), dtype=torch.float64)
    assert torch.equal(tot, total)


@pytest.mark.parametrize("n_a,E", [(30, 12), (10, 5), (100, 3)])
def test_rule_expert_actions_match_oracle(torch_cuda, n_a, E):
    """asm_batch_rule_actions_f32 (assembly.py:525-596) against the oracle's restatement on the device's own state,
    in the random and the packed regime, and over an episode driven by the expert itself (agents converge into
    the shapes: entering, exploration with the occupancy filter and sub-sampling, interaction all get used).
    Values: 1e-5 relative (device cos vs glibc, summation order) + one float32 rounding."""
    torch = torch_cuda
    import oracle_lib as ol
    from marl_amd.batched import BatchedAssemblyEnv
    from gpu_util import cfg_from_env
    env = BatchedAssemblyEnv(E, n_a, seed=40 + n_a, keep_indices=True)
    cfg = cfg_from_env(env)
    env.reset()
    rng = np.random.default_rng(n_a)
    grid = env.grid_center.cpu().numpy(); n_g = env.n_g.cpu().numpy(); l_cell = env.l_cell.cpu().numpy()
    p = env.p.cpu().numpy()
    for e in range(0, E, 2):
        p[e] = grid[e][:, rng.choice(n_g[e], n_a, replace=n_a > n_g[e])] + rng.normal(0, 0.04, (2, n_a))
    env.p.copy_(torch.from_numpy(p)); env.observe()
    clipped = inner = 0
    for t in range(12):
        a = env.rule_actions()
        got = a.cpu().numpy().astype(np.float64)
        P = env.p.cpu().numpy().astype(np.float64); V = env.dp.cpu().numpy().astype(np.float64)
        for e in range(E):
            g = np.ascontiguousarray(grid[e][:, :n_g[e]], dtype=np.float64)
            want = ol.rule_actions(cfg, np.ascontiguousarray(P[e]), np.ascontiguousarray(V[e]), g, float(l_cell[e]))
            np.testing.assert_allclose(got[e], want, rtol=1e-5, atol=2e-7, err_msg=f"n_a {n_a} step {t} env {e}")
            clipped += int((np.abs(want) == 1).sum()); inner += int((np.abs(want) < 1).sum())
        env.step(a)
    assert clipped > 0 and inner > 0
    plain = BatchedAssemblyEnv(2, n_a)
    plain.reset()
    with pytest.raises(RuntimeError, match="keep_indices"):
        plain.rule_actions()


@pytest.mark.parametrize("n_a,E", [(256, 5), (1024, 2), (100, 4)])
def test_large_swarm_step_statistics(torch_cuda, n_a, E):
    """The large-swarm kernel's in-step statistics (workgroup reductions, one-pass variance) against the metrics
    kernel on the same states, the contact pairs against the metrics of the state the step started from, and the
    reward / inside counts against the step's own outputs."""
    torch = torch_cuda
    from marl_amd.batched import BatchedAssemblyEnv
    env = BatchedAssemblyEnv(E, n_a, seed=n_a)
    env.reset()
    g = torch.Generator(device="cuda").manual_seed(1)
    cell = (torch.rand(E, n_a, device="cuda", generator=g) * env.n_g[:, None]).long()
    env.p.copy_(torch.gather(env.grid_center, 2, cell[:, None, :].expand(E, 2, n_a)) + 0.03 * torch.randn(E, 2, n_a, device="cuda", generator=g))
    env.observe()
    for t in range(3):
        before = env.stats.clone()
        pre = env.formation_metrics().cpu().numpy()
        _, rew, *_ = env.step(env.random_actions(step=t))
        d = (env.stats - before).cpu().numpy(); m = env.formation_metrics().cpu().numpy()
        np.testing.assert_allclose(d[:, 2], m[:, 0], rtol=1e-12)
        np.testing.assert_allclose(d[:, 3], m[:, 1], rtol=1e-5)
        np.testing.assert_allclose(d[:, 5], m[:, 4], rtol=1e-5)
        np.testing.assert_array_equal(d[:, 4], pre[:, 3])
        assert pre[:, 3].sum() > 0                                   # packed swarms do touch
        np.testing.assert_array_equal(d[:, 0], rew.sum(dim=(1, 2)).cpu().numpy())
        np.testing.assert_array_equal(d[:, 7], env.in_flags.sum(dim=1).cpu().numpy())
        assert (d[:, 6] == 1).all()
