"""GPU parity tests of the batched HIP path (C ABI group B of include/marl_amd_env.h)
against the CPU oracle and the golden vectors.  Run with -m gpu on an MI355X."""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROLLOUTS = sorted(glob.glob(os.path.join(GOLDEN, "rollout_*.npz")))


_RESET = []


def _schedule(monkeypatch, waves):
    """Select the step kernel's schedule for this test (asm_set_step_schedule); the fixture below restores the
    default afterwards.  (The MARL_AMD_FAST_T variable is read once per process, so it cannot switch schedules
    between tests.)"""
    from marl_amd import _lib
    lib = _lib.load()
    lib.asm_set_step_schedule(int(waves))
    _RESET.append(lib)


@pytest.fixture(autouse=True)
def _default_schedule_after_each_test():
    yield
    while _RESET:
        _RESET.pop().asm_set_step_schedule(-1)


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the GPU"
    return torch


def _env_from_golden(torch, z, E=1, keep_indices=True):
    from marl_amd.batched import BatchedAssemblyEnv
    grid = z["grid"]; n_g = grid.shape[1]
    env = BatchedAssemblyEnv(E, int(z["n_a"]), r_avoid=float(z["r_avoid"]), is_boundary=bool(z["is_boundary"]),
                             is_con_self_state=bool(z["is_con_self_state"]), d_sen=float(z["d_sen"]),
                             keep_indices=keep_indices, n_g_max=(n_g + 31) // 32 * 32)
    g = np.zeros((E, 2, env.n_g_max), np.float32); g[:, :, :n_g] = grid.astype(np.float32)
    env.grid_center.copy_(torch.from_numpy(g)); env.n_g.fill_(n_g); env.l_cell.fill_(float(z["l_cell"]))
    return env


def test_device_sqrt_f64_is_correctly_rounded(torch_cuda):
    """The kernels use device sqrt(double) for the in-shape test and the reward weights."""
    torch = torch_cuda
    import ctypes as C
    from marl_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(0, 30, 500000), rng.uniform(0, 1e-3, 200000), 10.0 ** rng.uniform(-12, 3, 300000),
                        np.array([0.0, 0.16, 0.4 ** 2, 0.0169, 0.0049])])
    xin = torch.from_numpy(x).cuda(); out = torch.empty_like(xin)
    _lib.check(lib.asm_debug_sqrt_f64(C.c_void_p(xin.data_ptr()), C.c_void_p(out.data_ptr()), x.size, None), "sqrt")
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), np.sqrt(x))


def test_device_cos_f32_error_budget(torch_cuda):
    """The reward's single-precision screen budgets 5e-7 for v_cos_f32 on [0, 0.5] revolutions
    (z / (2 d_sen) with z < d_sen); checked here over every float of that interval."""
    torch = torch_cuda
    import ctypes as C
    from marl_amd import _lib
    lib = _lib.load()
    out = torch.zeros(1, dtype=torch.float64, device="cuda")
    lo, hi = 0, int(np.float32(0.5).view(np.uint32))
    _lib.check(lib.asm_debug_cos_f32_max_err(lo, hi, C.c_void_p(out.data_ptr()), None), "cos")
    torch.cuda.synchronize()
    err = float(out.item())
    print("max |v_cos_f32 - cos| on [0, 0.5] rev:", err)
    assert 0.0 < err < 5e-7


@pytest.mark.parametrize("path", ROLLOUTS, ids=[os.path.basename(p)[8:-4] for p in ROLLOUTS])
def test_golden_states_observe_and_step(torch_cuda, path):
    """Every recorded reference state, rounded to the fp32 state the device holds:
    observation + teacher-forced step against the oracle (indices exact)."""
    torch = torch_cuda
    from gpu_util import cfg_from_env, check_observation, teacher_forced_step
    z = np.load(path)
    T = z["act"].shape[0]
    env = _env_from_golden(torch, z, E=T)
    cfg = cfg_from_env(env)
    # environment t of the batch = the recorded state before step t
    p = np.concatenate([z["p0"][None], z["p"][:-1]]).astype(np.float32)
    dp = np.concatenate([z["dp0"][None], z["dp"][:-1]]).astype(np.float32)
    env.p.copy_(torch.from_numpy(p)); env.dp.copy_(torch.from_numpy(dp))
    env.observe(want_reward=True)
    o = check_observation(env, cfg, True, "golden observe")
    # fp32 rounding of the state may flip a borderline decision w.r.t. the recorded
    # double-precision run, so the direct comparison is statistical, not exact
    nbr_ref = np.concatenate([z["nbr0"][None], z["nbr"][:-1]])
    assert (o["nbr"] == nbr_ref).mean() > 0.999
    teacher_forced_step(env, cfg, z["act"].astype(np.float32), "golden step")
    # sanity only: the recorded run started from the double-precision state, the device
    # from its float32 rounding, and a contact spring (k_ball = 30) amplifies that input
    # difference; the contract check is the teacher-forced one above
    np.testing.assert_allclose(env.p.cpu().numpy(), z["p"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(env.dp.cpu().numpy(), z["dp"], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("n_a,E,steps,boundary", [(30, 64, 25, True), (30, 33, 12, False), (10, 17, 10, True),
                                                   (7, 9, 6, True), (33, 5, 6, True), (64, 6, 6, True),
                                                   (100, 3, 4, True), (1, 4, 4, True)])
def test_random_rollout_teacher_forced(torch_cuda, n_a, E, steps, boundary):
    torch = torch_cuda
    from marl_amd.batched import BatchedAssemblyEnv
    from gpu_util import cfg_from_env, check_observation, teacher_forced_step
    env = BatchedAssemblyEnv(E, n_a, seed=100 + n_a, is_boundary=boundary, keep_indices=True)
    cfg = cfg_from_env(env)
    env.reset()
    check_observation(env, cfg, False, "reset")
    rng = np.random.default_rng(n_a)
    tot = 0.0
    for t in range(steps):
        act = rng.uniform(-1, 1, (E, 2, n_a)).astype(np.float32)
        o, _ = teacher_forced_step(env, cfg, act, f"step {t}")
        tot += o["rew"].sum()
    assert np.isfinite(tot)


@pytest.mark.parametrize("fast_t", ["0", "1", "2", "4"])
def test_kernel_schedules_against_oracle(torch_cuda, monkeypatch, fast_t):
    """The generic kernel (0) and the multi-wave kernel at 1, 2 and 4 waves per workgroup,
    each checked against the oracle in the packed-inside-the-shape regime."""
    torch = torch_cuda
    _schedule(monkeypatch, fast_t)
    from marl_amd.batched import BatchedAssemblyEnv
    from gpu_util import cfg_from_env, check_observation, teacher_forced_step
    E, n_a = 21, 30
    env = BatchedAssemblyEnv(E, n_a, seed=11, keep_indices=True)
    cfg = cfg_from_env(env)
    env.reset()
    rng = np.random.default_rng(4)
    grid = env.grid_center.cpu().numpy(); n_g = env.n_g.cpu().numpy()
    p = env.p.cpu().numpy()
    for e in range(0, E, 2):                                  # every other environment packed into its shape
        p[e] = grid[e][:, rng.choice(n_g[e], n_a, replace=False)] + rng.normal(0, 0.02, (2, n_a))
    env.p.copy_(torch.from_numpy(p)); env.observe()
    for t in range(6):
        teacher_forced_step(env, cfg, rng.uniform(-1, 1, (E, 2, n_a)).astype(np.float32), f"T={fast_t} step {t}")


def test_kernel_schedules_agree_bit_for_bit(torch_cuda, monkeypatch):
    """Same inputs through the generic kernel and the 1/2/4-wave kernels: identical outputs."""
    torch = torch_cuda
    from marl_amd.batched import BatchedAssemblyEnv
    E, n_a, steps = 301, 30, 12
    outs = {}
    for fast_t in ("0", "1", "2", "4"):
        _schedule(monkeypatch, fast_t)
        env = BatchedAssemblyEnv(E, n_a, seed=3, keep_indices=True)
        env.reset()
        assert env.step_schedule() == int(fast_t) + (16 if fast_t == "4" else 0)      # the kernel asked for is the one that runs
        rec = []
        for t in range(steps):
            obs, rew, _, _, prior = env.step(env.random_actions())
            rec.append([x.clone() for x in (obs, rew, prior, env.p, env.dp, env.neighbor_index, env.in_flags,
                                            env.nearest_cell, env.sensed_index, env.occupied_index)])
        outs[fast_t] = (rec, env.stats.clone())
    ref, ref_stats = outs["0"]
    for fast_t in ("1", "2", "4"):
        rec, st = outs[fast_t]
        for t in range(steps):
            for a, b in zip(rec[t], ref[t]):
                assert torch.equal(a, b), f"T={fast_t} step {t}"
        assert torch.allclose(st, ref_stats, rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("n_a,E", [(30, 301), (256, 3)])
def test_store_policies_agree_bit_for_bit(torch_cuda, monkeypatch, n_a, E):
    """Observation rows stored non-temporal (schedule + 64) or the ordinary way (+ 128): identical outputs, in the
    multi-wave kernel and in the large-swarm kernel."""
    torch = torch_cuda
    from marl_amd.batched import BatchedAssemblyEnv
    outs = []
    for bits in (64, 128):
        _schedule(monkeypatch, 4 + bits)
        env = BatchedAssemblyEnv(E, n_a, seed=5)
        env.reset()
        rec = []
        for t in range(6):
            obs, rew, _, _, prior = env.step(env.random_actions())
            rec.append([x.clone() for x in (obs, rew, prior, env.p, env.dp, env.neighbor_index, env.in_flags, env.nearest_cell)])
        outs.append((rec, env.stats.clone()))
    for t in range(6):
        for a, b in zip(outs[0][0][t], outs[1][0][t]):
            assert torch.equal(a, b), f"step {t}"
    assert torch.equal(outs[0][1], outs[1][1])


def test_converged_regime_teacher_forced(torch_cuda):
    """Agents packed inside the shape: sensed lists above 80, occupied cells, contacts,
    rewards — the regime random actions rarely reach."""
    torch = torch_cuda
    from marl_amd.batched import BatchedAssemblyEnv
    from gpu_util import cfg_from_env, check_observation, teacher_forced_step
    E, n_a = 48, 30
    env = BatchedAssemblyEnv(E, n_a, seed=5, keep_indices=True)
    cfg = cfg_from_env(env)
    env.reset()
    rng = np.random.default_rng(3)
    grid = env.grid_center.cpu().numpy(); n_g = env.n_g.cpu().numpy()
    p = np.zeros((E, 2, n_a), np.float32)
    for e in range(E):
        idx = rng.choice(n_g[e], n_a, replace=False)
        p[e] = grid[e][:, idx] + rng.normal(0, 0.03 if e % 2 else 0.004, (2, n_a))
    env.p.copy_(torch.from_numpy(p)); env.dp.copy_(torch.from_numpy(rng.uniform(-0.2, 0.2, (E, 2, n_a)).astype(np.float32)))
    env.observe(want_reward=True)
    o = check_observation(env, cfg, True, "packed observe")
    assert (o["sensed"] >= 0).sum(-1).max() == 80 and (o["occupied"] >= 0).sum(-1).max() > 20
    assert o["in_flags"].mean() > 0.8
    rew = 0; cont = 0
    for t in range(8):
        o, contacts = teacher_forced_step(env, cfg, rng.uniform(-0.3, 0.3, (E, 2, n_a)).astype(np.float32), f"packed {t}")
        rew += o["rew"].sum(); cont += contacts.sum()
    assert rew > 0 and cont > 0


def test_large_sensing_radius_subsamples_both_lists(torch_cuda):
    """d_sen = 1.2: occupied > 200 and sensed > 80, both subsampled (CPP:223-233, 243-261)."""
    torch = torch_cuda
    from gpu_util import cfg_from_env, check_observation
    z = np.load(os.path.join(GOLDEN, "fn_occ200.npz"))
    from marl_amd.batched import BatchedAssemblyEnv
    n_a = z["p"].shape[1]; n_g = z["grid"].shape[1]
    env = BatchedAssemblyEnv(2, n_a, r_avoid=float(z["r_avoid"]), d_sen=float(z["d_sen"]), keep_indices=True, n_g_max=544)
    g = np.zeros((2, 2, 544), np.float32); g[:, :, :n_g] = z["grid"].astype(np.float32)
    env.grid_center.copy_(torch.from_numpy(g)); env.n_g.fill_(n_g); env.l_cell.fill_(float(z["l_cell"]))
    env.p.copy_(torch.from_numpy(np.stack([z["p"], z["p"] + 0.01]).astype(np.float32)))
    env.dp.copy_(torch.from_numpy(np.stack([z["dp"], z["dp"]]).astype(np.float32)))
    env.observe(want_reward=True)
    o = check_observation(env, cfg_from_env(env), True, "occ200")
    assert (o["occupied"] >= 0).sum(-1).max() == 200 and (o["sensed"] >= 0).sum(-1).max() == 80


def test_reset_matches_oracle_and_shards(torch_cuda):
    torch = torch_cuda
    import oracle_lib as ol
    from marl_amd.batched import BatchedAssemblyEnv
    from gpu_util import ulp_diff_f32
    E, n_a = 96, 30
    env = BatchedAssemblyEnv(E, n_a, seed=226)
    env.reset(episode=3)
    sid, p, dp, grid, n_g, l_cell = ol.batch_reset(226, 3, 0, E, n_a, env.grid_center_origins, env.l_cells, env.n_g_max)
    assert np.array_equal(env.shape_id.cpu().numpy(), sid) and np.array_equal(env.n_g.cpu().numpy(), n_g)
    assert np.array_equal(env.l_cell.cpu().numpy(), l_cell)
    assert np.array_equal(env.p.cpu().numpy(), p.astype(np.float32))          # pure integer/fp64 arithmetic: exact
    assert np.array_equal(env.dp.cpu().numpy(), dp.astype(np.float32))
    # the grid goes through device cos/sin, which may differ from libm in the last bits
    assert ulp_diff_f32(env.grid_center.cpu().numpy(), grid).max() <= 1
    # the same global ids from a different shard give the same environments
    env2 = BatchedAssemblyEnv(40, n_a, seed=226, env_id_offset=56)
    env2.reset(episode=3)
    for name in ("p", "dp", "grid_center", "obs", "neighbor_index", "in_flags"):
        assert torch.equal(getattr(env2, name), getattr(env, name)[56:]), name


def test_full_size_properties(torch_cuda):
    """4096 x 30 (the benchmark configuration): properties that need no oracle pass
    over every environment, plus the oracle on a sample."""
    torch = torch_cuda
    from marl_amd.batched import BatchedAssemblyEnv
    from gpu_util import cfg_from_env
    import oracle_lib as ol
    E, n_a = 4096, 30
    env = BatchedAssemblyEnv(E, n_a, seed=226, keep_indices=True)
    env.reset()
    perm = torch.randperm(E, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    twin = BatchedAssemblyEnv(E, n_a, seed=226, keep_indices=True)
    for name in ("p", "dp", "grid_center", "n_g", "l_cell"):
        getattr(twin, name).copy_(getattr(env, name)[perm])
    twin.observe()
    rsum = 0.0
    for t in range(20):
        a = env.random_actions()
        obs, rew, done, _, prior = env.step(a)
        o2, r2, _, _, p2 = twin.step(a[perm].contiguous())
        # environments are independent: a permuted batch gives the permuted results, bit for bit
        assert torch.equal(o2, obs[perm]) and torch.equal(r2, rew[perm]) and torch.equal(p2, prior[perm])
        assert torch.equal(twin.neighbor_index, env.neighbor_index[perm]) and torch.equal(twin.sensed_index, env.sensed_index[perm])
        rsum += float(rew.sum())
        assert torch.isfinite(obs).all() and float(env.p.abs().max()) <= 2.4 + 0.1 and float(env.dp.abs().max()) <= float(np.float32(0.8))
        assert set(torch.unique(rew).tolist()) <= {0.0, 1.0}
        # sensed lists are ascending and -1 padded at the tail only
        s = env.sensed_index
        valid = s >= 0
        assert bool((valid[..., 1:] <= valid[..., :-1]).all())
        assert bool(((s[..., 1:] > s[..., :-1]) | ~valid[..., 1:]).all())
    st = env.episode_stats().cpu().numpy()
    assert st[0] == pytest.approx(rsum) and st[6] == 20 * E
    # oracle on a sample of environments of the final state
    cfg = cfg_from_env(env)
    idx = np.arange(0, E, 97)
    from gpu_util import oracle_observe_batch, assert_close
    o = oracle_observe_batch(cfg, env.p[idx].cpu().numpy(), env.dp[idx].cpu().numpy(), env.grid_center[idx].cpu().numpy(),
                             env.n_g[idx].cpu().numpy(), env.l_cell[idx].cpu().numpy())
    assert np.array_equal(env.neighbor_index[idx].cpu().numpy(), o["nbr"])
    assert np.array_equal(env.sensed_index[idx].cpu().numpy(), o["sensed"])
    assert np.array_equal(env.reward[idx].cpu().numpy().astype(np.float64), o["rew"])
    assert_close(env.obs[idx].cpu().numpy(), o["obs"], "obs sample")


def test_stats_match_wrapper_metrics(torch_cuda):
    """Per-step statistics against the oracle's restatement of the wrapper metrics."""
    torch = torch_cuda
    import oracle_lib as ol
    from marl_amd.batched import BatchedAssemblyEnv
    from gpu_util import cfg_from_env, oracle_dynamics_batch
    E, n_a = 24, 30
    env = BatchedAssemblyEnv(E, n_a, seed=9)
    cfg = cfg_from_env(env)
    env.reset()
    rng = np.random.default_rng(0)
    grid = env.grid_center.cpu().numpy(); n_g = env.n_g.cpu().numpy(); l_cell = env.l_cell.cpu().numpy()
    p = np.zeros((E, 2, n_a), np.float32)
    for e in range(E):
        p[e] = grid[e][:, rng.choice(n_g[e], n_a, replace=False)] + rng.normal(0, 0.02, (2, n_a))
    env.p.copy_(torch.from_numpy(p)); env.observe()
    want = np.zeros((E, 8))
    for t in range(3):
        act = rng.uniform(-1, 1, (E, 2, n_a)).astype(np.float32)
        p0 = env.p.cpu().numpy(); dp0 = env.dp.cpu().numpy(); nbr0 = env.neighbor_index.cpu().numpy()
        _, _, _, contacts = oracle_dynamics_batch(cfg, p0, dp0, act, grid, n_g, l_cell, nbr0)
        obs, rew, *_ = env.step(torch.from_numpy(act).cuda())
        p1 = env.p.cpu().numpy().astype(np.float64); r = rew.cpu().numpy().astype(np.float64)
        for e in range(E):
            g = np.ascontiguousarray(grid[e][:, :n_g[e]], dtype=np.float64)
            m = ol.metrics(np.ascontiguousarray(p1[e]), g, env.r_avoid)
            want[e] += [r[e].sum(), (r[e] ** 2).sum(), m["coverage_rate"], m["distribution_uniformity"], contacts[e],
                        m["avg_distance_to_target"], 1.0, env.in_flags[e].sum().item()]
    # the nearest-neighbour distance of the uniformity metric is picked by an fp32 screen
    np.testing.assert_allclose(env.stats.cpu().numpy(), want, rtol=1e-5, atol=1e-9)


def test_big_swarm_one_step(torch_cuda):
    """256-agent and 1024-agent environments (one workgroup per environment)."""
    torch = torch_cuda
    from marl_amd.batched import BatchedAssemblyEnv
    from gpu_util import cfg_from_env, check_observation, teacher_forced_step
    for n_a, E in ((256, 3), (1024, 2)):
        env = BatchedAssemblyEnv(E, n_a, seed=n_a, keep_indices=True)
        cfg = cfg_from_env(env)
        env.reset()
        check_observation(env, cfg, False, f"reset {n_a}")
        rng = np.random.default_rng(n_a)
        for t in range(2):
            teacher_forced_step(env, cfg, rng.uniform(-1, 1, (E, 2, n_a)).astype(np.float32), f"n_a {n_a} step {t}")


def test_argument_errors(torch_cuda):
    from marl_amd.batched import BatchedAssemblyEnv
    with pytest.raises(RuntimeError, match="n_a"):
        BatchedAssemblyEnv(2, 2000).reset()
    env = BatchedAssemblyEnv(2, 30)
    env.reset()
    import torch
    with pytest.raises(ValueError):
        env.step(torch.zeros(2, 2, 29, device="cuda"))


def test_stream_shards_are_the_same_environments(torch_cuda):
    """Two shards on two HIP streams (marl_amd.sharding.StreamShards) against one batch of the same global ids."""
    torch = torch_cuda
    from marl_amd.batched import BatchedAssemblyEnv
    from marl_amd.sharding import StreamShards
    E, n_a = 37, 30
    one = BatchedAssemblyEnv(E, n_a, seed=9, env_id_offset=100)
    two = StreamShards(E, n_a, parts=2, seed=9, env_id_offset=100)
    assert [e.E for e in two.envs] == [19, 18] and [e.env_id_offset for e in two.envs] == [100, 119]
    o1 = one.reset()
    o2 = two.reset()
    two.synchronize()
    assert torch.equal(o1, torch.cat(o2))
    for t in range(7):
        a = one.random_actions(step=t)
        parts = two.on_each(lambda k, env: env.random_actions(step=t))
        two.synchronize()
        assert torch.equal(a, torch.cat(parts))
        r1 = one.step(a)
        r2 = two.step(parts)
        two.synchronize()
        for j in (0, 1, 4):
            assert torch.equal(r1[j], torch.cat([r[j] for r in r2]))
    assert torch.allclose(one.episode_stats(), two.episode_stats(), rtol=1e-12, atol=0)


def test_nearest_cell_exact_ties(torch_cuda):
    """Agents exactly half way between two cells of a row (and at the centre of four): the distances tie in
    double precision and the reference keeps the first minimum (np.argmin / AssemblyEnv.cpp:889)."""
    torch = torch_cuda
    from marl_amd.batched import BatchedAssemblyEnv
    from gpu_util import cfg_from_env, check_observation
    E, n_a = 7, 30
    env = BatchedAssemblyEnv(E, n_a, seed=1, keep_indices=True)
    cfg = cfg_from_env(env)
    env.reset()
    rng = np.random.default_rng(8)
    p = env.p.cpu().numpy()
    n_tied = 0
    for e in range(E):
        env.set_shape(e, torch.tensor([e]))
        g = env.grid_center[e].cpu().numpy()[:, :env.n_gs[e]].astype(np.float64)
        found = []
        for c in rng.permutation(env.n_gs[e] - 1):
            for d in (1, 2):                                    # neighbour in the row, and the one after it
                if c + d >= env.n_gs[e] or g[1, c] != g[1, c + d]:
                    continue
                mx = np.float32((g[0, c] + g[0, c + d]) / 2)
                for my in (np.float32(g[1, c]), np.float32(g[1, c] + 0.01)):
                    s = (g[0] - float(mx)) ** 2 + (g[1] - float(my)) ** 2
                    if s[c] == s[c + d] and (d == 1 and s.min() == s[c]):
                        found.append((mx, my, c))
            if len(found) >= n_a:
                break
        assert len(found) >= n_a // 2, "shape offers too few exactly representable midpoints"
        for i, (mx, my, c) in enumerate(found[:n_a]):
            p[e, 0, i] = mx; p[e, 1, i] = my
        n_tied += min(len(found), n_a)
    env.p.copy_(torch.from_numpy(p))
    env.observe(want_reward=True)
    check_observation(env, cfg, True, "ties")
    near = env.nearest_cell.cpu().numpy()
    pd = env.p.cpu().numpy().astype(np.float64); gd = env.grid_center.cpu().numpy().astype(np.float64)
    for e in range(E):
        n_g = env.n_gs[e]
        s = (gd[e, 0, None, :n_g] - pd[e, 0, :, None]) ** 2 + (gd[e, 1, None, :n_g] - pd[e, 1, :, None]) ** 2
        assert np.array_equal(near[e], np.argmin(np.sqrt(s), axis=1)), f"env {e}"
        ties = (np.sort(s, axis=1)[:, 0] == np.sort(s, axis=1)[:, 1]).sum()
        assert ties >= n_a // 2
    # the step kernel on the same state (zero action, zero velocity: the agents stay put)
    env.dp.zero_()
    env.step(torch.zeros(E, 2, n_a, device="cuda"))
    moved = np.abs(env.p.cpu().numpy() - p).max()
    if moved == 0:
        assert np.array_equal(env.nearest_cell.cpu().numpy(), near)


def test_equidistant_neighbours_keep_index_order(torch_cuda):
    """Two neighbours at exactly the same distance, and a closer one with a higher index that is inserted after
    them: the list stays ordered by (squared distance, index).  (The reference's std::sort leaves the order
    of equal norms unspecified; the oracle defines it.)"""
    torch = torch_cuda
    from marl_amd.batched import BatchedAssemblyEnv
    from gpu_util import cfg_from_env, check_observation
    n_a = 30
    env = BatchedAssemblyEnv(3, n_a, seed=1, keep_indices=True)
    cfg = cfg_from_env(env)
    env.reset()
    p = np.zeros((2, n_a)); p[0] = np.linspace(-2.3, 2.3, n_a); p[1] = 2.0
    P = np.array([np.float32(-0.74249387), np.float32(0.27645427)], dtype=np.float64)
    d = float(np.float32(0.27645427) - np.float32(0.21534318))
    p[:, 25] = P; p[:, 11] = P + [0, -d]; p[:, 19] = P + [-d, 0]; p[:, 24] = P + [0, -0.01]
    p[:, 3] = P + [d, 0]; p[:, 28] = P + [0, d]; p[:, 7] = P + [0.02, 0]; p[:, 29] = P + [-0.02, 0]
    p = p.astype(np.float32)
    env.p.copy_(torch.from_numpy(np.stack([p, p[:, ::-1].copy(), p])))
    env.dp.zero_()
    env.observe(want_reward=True)
    check_observation(env, cfg, True, "equidistant observe")
    assert env.neighbor_index[0, 25].cpu().tolist() == [24, 7, 29, 3, 11, 19]
    env.step(torch.zeros(3, 2, n_a, device="cuda"))            # agents that touch are pushed; the check is against the oracle
    check_observation(env, cfg, True, "equidistant step")


def test_captured_steps_equal_eager_steps(torch_cuda):
    """StreamShards.capture: a HIP graph of several steps over the shard streams leaves exactly the state,
    observations and statistics of the same steps launched one by one; replays and eager steps can alternate."""
    torch = torch_cuda
    from marl_amd.sharding import StreamShards
    E, n_a, G = 50, 30, 6
    a_sh = StreamShards(E, n_a, parts=2, seed=4)
    b_sh = StreamShards(E, n_a, parts=2, seed=4)
    a_sh.reset(0); b_sh.reset(0)
    acts = a_sh.on_each(lambda k, env: torch.stack([env.random_actions(step=j) for j in range(G)]))
    a_sh.synchronize()                                                       # b_sh reads them on streams of its own
    a_sh.step([a[0] for a in acts]); b_sh.step([a[0] for a in acts])          # kernels loaded before the capture
    graph = a_sh.capture(lambda: [a_sh.step([a[j] for a in acts]) for j in range(G)])
    assert a_sh.envs[0].step_count == 1                                      # capturing ran nothing
    for rnd in range(2):
        graph.replay()
        a_sh.step([a[1] for a in acts])                                      # eager work queued behind the replay
        for j in range(G):
            b_sh.step([a[j] for a in acts])
        b_sh.step([a[1] for a in acts])
    a_sh.synchronize(); b_sh.synchronize(); torch.cuda.synchronize()
    assert a_sh.envs[0].step_count == b_sh.envs[0].step_count == 1 + 2 * (G + 1)
    for ea, eb in zip(a_sh.envs, b_sh.envs):
        for name in ("p", "dp", "obs", "reward", "a_prior", "neighbor_index", "in_flags", "stats"):
            assert torch.equal(getattr(ea, name), getattr(eb, name)), name
    assert torch.equal(a_sh.episode_stats(), b_sh.episode_stats())


@pytest.mark.parametrize("n_a,step_denominator,fast_t", [(30, 64, "4"), (30, 16, "4"), (16, 32, "4"), (100, 32, "4"),
                                                           (30, 64, "0"), (30, 16, "1"), (16, 32, "2")])
def test_lattice_states_full_of_ties(torch_cuda, monkeypatch, n_a, step_denominator, fast_t):
    """Agents snapped to a coarse binary lattice (distinct points): equal neighbour distances, equal distances to
    cells of the unrotated shape and distances that hit thresholds' neighbourhoods are the rule here, not the
    exception.  Observation and teacher-forced steps against the oracle, indices exact (fast_t: the kernel
    schedule, "0" = the generic kernel)."""
    torch = torch_cuda
    _schedule(monkeypatch, fast_t)
    from marl_amd.batched import BatchedAssemblyEnv
    from gpu_util import cfg_from_env, check_observation, teacher_forced_step
    E = 12
    env = BatchedAssemblyEnv(E, n_a, seed=31, keep_indices=True)
    cfg = cfg_from_env(env)
    env.reset()
    for e in range(0, E, 2):
        env.set_shape(e % env.num_train_shape, torch.tensor([e]))           # unrotated, centred: symmetric cell rows
    rng = np.random.default_rng(step_denominator + n_a)
    q = 1.0 / step_denominator
    p = np.zeros((E, 2, n_a), np.float32)
    for e in range(E):
        half = 0.5 if e % 3 else 1.5                                        # dense clusters and spread-out swarms
        m = int(round(half / q))
        side = 2 * m + 1
        pick = rng.choice(side * side, n_a, replace=False)                  # distinct lattice points
        p[e, 0] = (pick % side - m) * q; p[e, 1] = (pick // side - m) * q
    env.p.copy_(torch.from_numpy(p))
    v = rng.integers(-2, 3, (E, 2, n_a)).astype(np.float32) * 0.25          # velocities on a lattice too
    env.dp.copy_(torch.from_numpy(v))
    env.observe(want_reward=True)
    check_observation(env, cfg, True, "lattice observe")
    for t in range(3):
        act = rng.integers(-4, 5, (E, 2, n_a)).astype(np.float32) * 0.25
        teacher_forced_step(env, cfg, act, f"lattice step {t}")


@pytest.mark.parametrize("n_g_max", [None, 1056])
def test_extreme_grid_sizes(torch_cuda, n_g_max):
    """Shapes of 1024 cells (32 mask words, the largest the multi-wave kernel takes), 33 cells, 5 cells and a
    single cell in one batch; with n_g_max = 1056 the same batch goes through the generic kernel."""
    torch = torch_cuda
    from marl_amd.batched import BatchedAssemblyEnv
    from gpu_util import cfg_from_env, check_observation, teacher_forced_step
    l = 0.05
    yy, xx = np.mgrid[0:32, 0:32]
    big = np.stack([(xx.ravel() - 15.5) * l, (yy.ravel() - 15.5) * l], axis=1)          # 1024 cells
    shapes = ([l, l, l, l], [big, big[:33] + [0.0, 0.8], big[:5] + [0.4, 0.8], big[:1] + [0.8, 0.8]])
    E, n_a = 9, 30
    env = BatchedAssemblyEnv(E, n_a, seed=77, shapes=shapes, r_avoid=0.26, keep_indices=True, n_g_max=n_g_max)
    assert env.n_g_max == (1024 if n_g_max is None else 1056)
    cfg = cfg_from_env(env)
    env.reset()
    assert set(env.n_g.cpu().tolist()) <= {1024, 33, 5, 1}
    check_observation(env, cfg, False, "reset")
    rng = np.random.default_rng(1)
    grid = env.grid_center.cpu().numpy(); n_g = env.n_g.cpu().numpy(); p = env.p.cpu().numpy()
    for e in range(0, E, 2):                                    # half of the environments: agents onto their shape
        p[e] = grid[e][:, rng.integers(0, n_g[e], n_a)] + rng.normal(0, 0.05, (2, n_a))
    env.p.copy_(torch.from_numpy(p)); env.observe(want_reward=True)
    check_observation(env, cfg, True, "packed observe")
    for t in range(3):
        teacher_forced_step(env, cfg, rng.uniform(-1, 1, (E, 2, n_a)).astype(np.float32), f"step {t}")
    assert (env.sensed_index >= 0).sum(dim=2).max().item() == 80     # the big shape fills (and sub-samples) the list


@pytest.mark.parametrize("cap,occ_cap,fast_t", [(80, 200, "4"), (30, 40, "4"), (100, 64, "4"), (6, 8, "4"), (30, 40, "2"),
                                                (100, 64, "1"), (31, 40, "4"), (30, 40, "0")])
def test_other_list_capacities(torch_cuda, monkeypatch, cap, occ_cap, fast_t):
    """num_obs_grid_max / num_occupied_grid_max other than the reference's 80 / 200 (assembly.py:126-128): slot
    ranges per wave, sub-sampling and the observation layout all follow the capacity.  An odd capacity is served
    by the generic kernel."""
    torch = torch_cuda
    _schedule(monkeypatch, fast_t)
    from marl_amd.batched import BatchedAssemblyEnv
    from gpu_util import cfg_from_env, check_observation, teacher_forced_step
    E, n_a = 10, 30
    env = BatchedAssemblyEnv(E, n_a, seed=cap, keep_indices=True, num_obs_grid_max=cap, num_occupied_grid_max=occ_cap)
    assert env.obs.shape[1] == 32 + 2 * cap
    cfg = cfg_from_env(env)
    env.reset()
    rng = np.random.default_rng(cap)
    grid = env.grid_center.cpu().numpy(); n_g = env.n_g.cpu().numpy(); p = env.p.cpu().numpy()
    for e in range(0, E, 2):
        p[e] = grid[e][:, rng.choice(n_g[e], n_a, replace=False)] + rng.normal(0, 0.03, (2, n_a))
    env.p.copy_(torch.from_numpy(p)); env.observe(want_reward=True)
    check_observation(env, cfg, True, "observe")
    for t in range(3):
        teacher_forced_step(env, cfg, rng.uniform(-1, 1, (E, 2, n_a)).astype(np.float32), f"cap {cap} step {t}")


@pytest.mark.parametrize("fast_t", ["4", "0"])
def test_non_default_physics(torch_cuda, monkeypatch, fast_t):
    """Every scalar of asm_params away from the reference's default (mass != 1 takes the division of
    assembly.py:638): dynamics, prior and observation against the oracle configured alike."""
    torch = torch_cuda
    _schedule(monkeypatch, fast_t)
    from marl_amd.batched import BatchedAssemblyEnv
    from gpu_util import cfg_from_env, check_observation, teacher_forced_step
    E, n_a = 8, 30
    env = BatchedAssemblyEnv(E, n_a, seed=3, keep_indices=True, r_avoid=0.31, d_sen=0.55)
    over = dict(size_a=0.05, vel_max=0.5, k_ball=22.0, k_wall=70.0, c_wall=3.5, dt=0.07, sensitivity=0.6, mass=2.5,
                half_w=2.0, half_h=2.2)
    for k, v in over.items():
        setattr(env.params, k, v)
    cfg = cfg_from_env(env)
    for k, v in over.items():
        setattr(cfg, "Vel_max" if k == "vel_max" else k, v)
    env.reset()
    rng = np.random.default_rng(0)
    p = env.p.cpu().numpy()
    p[:, 0] = np.clip(p[:, 0], -2.05, 2.05); p[:, 1] = np.clip(p[:, 1], -2.25, 2.25)     # some agents into the (moved) walls
    for e in range(0, E, 2):
        grid = env.grid_center[e].cpu().numpy(); n_g = int(env.n_g[e])
        p[e] = grid[:, rng.choice(n_g, n_a, replace=False)] + rng.normal(0, 0.03, (2, n_a))
    env.p.copy_(torch.from_numpy(p.astype(np.float32))); env.observe(want_reward=True)
    check_observation(env, cfg, True, "observe")
    for t in range(4):
        teacher_forced_step(env, cfg, rng.uniform(-1, 1, (E, 2, n_a)).astype(np.float32), f"physics step {t}")


def test_without_prior_and_without_self_state(torch_cuda):
    """training_method other than 'llm_rl' (no prior, assembly.py:600) and is_con_self_state = False (the
    observation drops the agent's own four rows, assembly.py:105, 806)."""
    torch = torch_cuda
    from marl_amd.batched import BatchedAssemblyEnv
    from gpu_util import cfg_from_env, check_observation, teacher_forced_step
    E, n_a = 7, 30
    env = BatchedAssemblyEnv(E, n_a, seed=13, keep_indices=True, training_method="rl", is_con_self_state=False)
    assert env.obs.shape[1] == 188
    cfg = cfg_from_env(env)
    env.reset()
    check_observation(env, cfg, False, "reset")
    rng = np.random.default_rng(6)
    for t in range(3):
        act = rng.uniform(-1, 1, (E, 2, n_a)).astype(np.float32)
        teacher_forced_step(env, cfg, act, f"step {t}")
    assert env.step(torch.zeros(E, 2, n_a, device="cuda"))[4] is None


def test_periodic_arena_kernels_agree(torch_cuda, monkeypatch):
    """is_boundary = False (assembly.py:97-101): the multi-wave kernel's periodic instantiation against the generic
    kernel, bit for bit, over an episode start in which agents cross the arena's edges (wrapped positions, wrapped
    neighbour offsets, agent 0's wrapped contact row in the statistics)."""
    torch = torch_cuda
    from marl_amd.batched import BatchedAssemblyEnv
    E, n_a, steps = 64, 30, 40
    outs = {}
    for mode in ("4", "0"):
        _schedule(monkeypatch, mode)
        env = BatchedAssemblyEnv(E, n_a, seed=91, is_boundary=False, keep_indices=True)
        env.reset()
        p = env.p.clone()
        p[::2] = p[::2] * 0 + torch.tensor([2.3, -2.3], device="cuda")[None, :, None] + 0.15 * torch.randn(
            E // 2, 2, n_a, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
        p.clamp_(-2.4, 2.4)
        env.p.copy_(p); env.dp.mul_(1.6).clamp_(-0.8, 0.8); env.observe()
        rec = []
        for t in range(steps):
            obs, rew, _, _, prior = env.step(env.random_actions())
            rec.append([x.clone() for x in (obs, rew, prior, env.p, env.dp, env.neighbor_index, env.in_flags, env.sensed_index)])
        outs[mode] = (rec, env.stats.clone())
        crossed = (env.p.abs() > 2.3).any().item()
    for t in range(steps):
        for k, (a, b) in enumerate(zip(outs["4"][0][t], outs["0"][0][t])):
            assert torch.equal(a, b), f"step {t} output {k}"
    assert torch.allclose(outs["4"][1], outs["0"][1], rtol=1e-5, atol=1e-9)
    assert crossed
    assert (outs["4"][0][-1][3].abs() <= 2.4).all()


@pytest.mark.parametrize("boundary", [True, False])
def test_index_list_levels_agree(torch_cuda, boundary):
    """The step kernel is compiled once per index-list level (none / sensed_index / both).  The oracle comparisons
    above run with both lists; here the three builds step the same episode, half of the environments packed into
    their shapes (occupancy filter, sub-sampling), and every other output must be bit-equal, so that parity carries
    over to the build without lists that the throughput path uses."""
    torch = torch_cuda
    from marl_amd.batched import BatchedAssemblyEnv
    E, n_a, steps = 48, 30, 25
    runs = {}
    for level in (True, "sensed", False):
        env = BatchedAssemblyEnv(E, n_a, seed=61, is_boundary=boundary, keep_indices=level)
        env.reset()
        g = torch.Generator(device="cuda").manual_seed(9)
        cell = (torch.rand(E, n_a, device="cuda", generator=g) * env.n_g[:, None]).long()
        pos = torch.gather(env.grid_center, 2, cell[:, None, :].expand(E, 2, n_a)) + 0.03 * torch.randn(E, 2, n_a, device="cuda", generator=g)
        p = env.p.clone(); p[::2] = pos[::2]
        env.p.copy_(p); env.observe()
        rec = [[x.clone() for x in (env.obs, env.neighbor_index, env.in_flags)]]
        for t in range(steps):
            obs, rew, _, _, prior = env.step(env.random_actions())
            rec.append([x.clone() for x in (obs, rew, prior, env.p, env.dp, env.neighbor_index, env.in_flags, env.stats)])
            if level:
                rec[-1].append(env.sensed_index.clone())
        runs[level] = rec
        if level is True:
            assert (env.in_flags[::2].sum() > 0) and (env.occupied_index >= 0).any() and (env.sensed_index[..., -1] >= 0).any()
    for t in range(steps + 1):
        for k in range(len(runs[False][t])):
            assert torch.equal(runs[True][t][k], runs[False][t][k]), f"none vs both: step {t} output {k}"
            assert torch.equal(runs[True][t][k], runs["sensed"][t][k]), f"sensed vs both: step {t} output {k}"
        if t:
            assert torch.equal(runs[True][t][-1], runs["sensed"][t][-1]), f"sensed_index: step {t}"


def test_index_list_levels_agree_full_size(torch_cuda):
    """The same at the benchmark's size (4096 environments x 30 agents, two shards of 2048 on two streams as the
    bench steps them): the build without lists against the build with both, every step's observation, reward and
    prior bit-equal, and the final state and statistics."""
    torch = torch_cuda
    from marl_amd.sharding import StreamShards
    E, n_a, steps = 4096, 30, 30
    runs = {}
    for level in (True, False):
        shards = StreamShards(E, n_a, parts=2, seed=226, keep_indices=level)
        shards.reset(0)
        acts = shards.on_each(lambda k, env: [env.random_actions(step=t).clone() for t in range(steps)])
        rec = []
        for t in range(steps):
            shards.step([a[t] for a in acts])
            shards.synchronize()
            rec.append([torch.cat([getattr(env, name) for env in shards.envs]) for name in ("obs", "reward", "a_prior")])
            torch.cuda.synchronize()                              # the copies ran on the default stream
        final = [torch.cat([getattr(env, name) for env in shards.envs])
                 for name in ("p", "dp", "neighbor_index", "in_flags", "nearest_cell", "stats")]
        torch.cuda.synchronize()
        runs[level] = (rec, final)
    for t in range(steps):
        for k in range(3):
            assert torch.equal(runs[True][0][t][k], runs[False][0][t][k]), f"step {t} output {k}"
    for k in range(6):
        assert torch.equal(runs[True][1][k], runs[False][1][k]), f"final {k}"
    assert runs[False][1][3].sum() > 0                             # some agents did get inside their shapes
