"""The lattice path of the step kernel (asm_lattice, include/marl_amd_env.h) against the all-cells path and the
oracle: same environments stepped with and without the description must agree bit for bit in every output, in
the random regime, packed inside the shapes, after shape switches, and with agents placed exactly on lattice
sites, half-way points and the sensing circle's lattice crossings (the undecided band)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
NAMES = ("obs", "reward", "a_prior", "p", "dp", "neighbor_index", "in_flags", "nearest_cell", "sensed_index", "occupied_index")


def _pair(E, n_a=30, seed=3, **kw):
    from marl_amd.batched import BatchedAssemblyEnv
    a = BatchedAssemblyEnv(E, n_a, seed=seed, keep_indices=True, **kw)
    b = BatchedAssemblyEnv(E, n_a, seed=seed, keep_indices=True, use_lattice=False, **kw)
    assert a._lattice is not None and b._lattice is None
    return a, b


def _same(torch, a, b, tag):
    for name in NAMES:
        assert torch.equal(getattr(a, name), getattr(b, name)), f"{tag}: {name}"


def test_lattice_description_matches_reset_grids(torch_cuda):
    a, _ = _pair(257)
    a.reset()
    assert a.lattice_valid and a.check_lattice(1e-6) == 0
    a.grid_center[5, 0, 17] += 1e-3
    assert a.check_lattice(1e-6) == 1
    a.set_shape(4)
    assert a.check_lattice(1e-6) == 0


def test_lattice_path_equals_all_cells_path_random_regime(torch_cuda):
    torch = torch_cuda
    a, b = _pair(1501)
    a.reset(); b.reset()
    assert a.lattice_valid and not b.lattice_valid
    assert a.step_schedule() == 4 + 16 and b.step_schedule() == 4
    _same(torch, a, b, "reset")
    for t in range(40):
        act = a.random_actions(step=t)
        a.step(act); b.step(act)
        _same(torch, a, b, f"step {t}")
    assert torch.allclose(a.stats, b.stats, rtol=1e-12, atol=0)


def test_lattice_path_equals_all_cells_path_packed_and_oracle(torch_cuda):
    torch = torch_cuda
    from gpu_util import cfg_from_env, teacher_forced_step
    E, n_a = 96, 30
    a, b = _pair(E, seed=5)
    a.reset(); b.reset()
    rng = np.random.default_rng(3)
    grid = a.grid_center.cpu().numpy(); n_g = a.n_g.cpu().numpy()
    p = np.zeros((E, 2, n_a), np.float32)
    for e in range(E):
        idx = rng.choice(n_g[e], n_a, replace=False)
        p[e] = grid[e][:, idx] + rng.normal(0, (0.06, 0.02, 0.004)[e % 3], (2, n_a))
    for env in (a, b):
        env.p.copy_(torch.from_numpy(p)); env.observe(want_reward=True)
    _same(torch, a, b, "packed observe")
    assert int((a.sensed_index >= 0).sum(-1).max()) == 80 and int((a.occupied_index >= 0).sum(-1).max()) > 20
    cfg = cfg_from_env(a)
    for t in range(10):
        act = rng.uniform(-0.4, 0.4, (E, 2, n_a)).astype(np.float32)
        teacher_forced_step(a, cfg, act, f"lattice packed {t}")          # lattice path == oracle
        b.step(torch.from_numpy(act).cuda())
        _same(torch, a, b, f"packed step {t}")
    assert torch.allclose(a.stats, b.stats, rtol=1e-12, atol=0)


def test_lattice_path_on_exact_lattice_positions(torch_cuda):
    """Unrotated, centred shapes (set_shape): the cell coordinates are the float-rounded lattice, and agents are put
    exactly on cells, half-way between cells, at cell corners and at multiples of the pitch away from cells where the
    sensing circle passes through lattice sites (d_sen = 6 pitches for one shape): ties and band cases everywhere."""
    torch = torch_cuda
    from gpu_util import cfg_from_env, check_observation
    E, n_a = 42, 30
    a, b = _pair(E, seed=9)
    a.reset(); b.reset()
    rng = np.random.default_rng(11)
    p = np.zeros((E, 2, n_a), np.float64)
    for e in range(E):
        k = e % 7
        for env in (a, b):
            env.set_shape(k, torch.tensor([e]))
        g = a.grid_center_origins[k]; l = a.l_cells[k]
        idx = rng.choice(g.shape[0], n_a, replace=False)
        kind = rng.integers(0, 6, n_a)
        off = np.zeros((n_a, 2))
        off[kind == 1] = [0.5 * l, 0.0]                                  # between two cells of a row
        off[kind == 2] = [0.5 * l, 0.5 * l]                              # centre of four
        off[kind == 3] = np.stack([rng.integers(-9, 10, (kind == 3).sum()) * l, rng.integers(-9, 10, (kind == 3).sum()) * l], 1)
        off[kind == 4] = rng.normal(0, 1e-7, ((kind == 4).sum(), 2))     # rounding distance from a cell
        off[kind == 5] = rng.uniform(-3, 3, ((kind == 5).sum(), 2)) * l
        q = np.clip(g[idx] + off, -2.3, 2.3)
        _, first = np.unique(q.astype(np.float32), axis=0, return_index=True)   # coincident agents are outside the contract
        dup = np.setdiff1d(np.arange(n_a), first)
        q[dup] = g[idx[dup]] + rng.uniform(-0.3, 0.3, (len(dup), 2)) * l
        p[e] = q.T
    a.lattice_valid = True                                               # every environment was switched
    for env in (a, b):
        env.p.copy_(torch.from_numpy(p.astype(np.float32))); env.dp.zero_(); env.observe(want_reward=True)
    assert a.check_lattice(1e-6) == 0
    _same(torch, a, b, "lattice positions")
    check_observation(a, cfg_from_env(a), True, "lattice positions vs oracle")
    for t in range(4):
        act = a.random_actions(step=t) * 0.05
        a.step(act); b.step(act)
        _same(torch, a, b, f"lattice positions step {t}")


def test_lattice_path_full_size_and_off(torch_cuda):
    """4096 environments, 30 steps: both paths bit-equal; invalidate_lattice() falls back to the all-cells kernel."""
    torch = torch_cuda
    a, b = _pair(4096, seed=226)
    a.reset(); b.reset()
    for t in range(30):
        act = a.random_actions(step=t)
        a.step(act); b.step(act)
        if t % 5 == 4:
            _same(torch, a, b, f"full size step {t}")
    a.invalidate_lattice()
    assert a._lat() is None
    act = a.random_actions(step=99)
    a.step(act); b.step(act)
    _same(torch, a, b, "after invalidate")


@pytest.mark.parametrize("n_a,E", [(100, 9), (256, 5), (1024, 2)])
def test_lattice_path_large_swarms(torch_cuda, n_a, E):
    """The large-swarm schedule (a lane per agent, one environment per workgroup) with and without the lattice
    description: bit-equal, random and packed, and equal to the oracle."""
    torch = torch_cuda
    from gpu_util import cfg_from_env, teacher_forced_step
    a, b = _pair(E, n_a=n_a, seed=n_a)
    a.reset(); b.reset()
    assert a.step_schedule() & 63 == 32 and b.step_schedule() & 63 == 0     # (+ 64: pair passes through cell lists)
    for t in range(3):
        act = a.random_actions(step=t)
        a.step(act); b.step(act)
        _same(torch, a, b, f"swarm random {t}")
    g = torch.Generator(device="cuda").manual_seed(2)
    cell = (torch.rand(E, n_a, device="cuda", generator=g) * a.n_g[:, None]).long()
    pos = torch.gather(a.grid_center, 2, cell[:, None, :].expand(E, 2, n_a)) + 0.03 * torch.randn(E, 2, n_a, device="cuda", generator=g)
    for env in (a, b):
        env.p.copy_(pos); env.observe()
    cfg = cfg_from_env(a)
    rng = np.random.default_rng(1)
    for t in range(2):
        act = rng.uniform(-1, 1, (E, 2, n_a)).astype(np.float32)
        teacher_forced_step(a, cfg, act, f"swarm packed {t}")
        b.step(torch.from_numpy(act).cuda())
        _same(torch, a, b, f"swarm packed {t}")
    assert torch.allclose(a.stats, b.stats, rtol=1e-12, atol=0)


def test_lattice_path_periodic_arena(torch_cuda):
    """is_boundary=False (no walls, wrapped positions and neighbour offsets; the cell computations are never wrapped,
    AssemblyEnv.cpp:93-95 vs 863-913): lattice path on, against the all-cells path and the oracle."""
    torch = torch_cuda
    from gpu_util import cfg_from_env, teacher_forced_step
    E, n_a = 65, 30
    a, b = _pair(E, seed=21, is_boundary=False)
    a.reset(); b.reset()
    assert a.step_schedule() == 4 + 16 and b.step_schedule() == 4
    cfg = cfg_from_env(a)
    rng = np.random.default_rng(2)
    for t in range(12):
        act = rng.uniform(-1, 1, (E, 2, n_a)).astype(np.float32)
        teacher_forced_step(a, cfg, act, f"periodic lattice {t}")
        b.step(torch.from_numpy(act).cuda())
        _same(torch, a, b, f"periodic {t}")
    grid = a.grid_center.cpu().numpy(); n_g = a.n_g.cpu().numpy()
    p = np.stack([grid[e][:, rng.choice(n_g[e], n_a, replace=False)] + rng.normal(0, 0.02, (2, n_a)) for e in range(E)]).astype(np.float32)
    for env in (a, b):
        env.p.copy_(torch.from_numpy(p)); env.observe()
    for t in range(6):
        act = rng.uniform(-1, 1, (E, 2, n_a)).astype(np.float32)
        teacher_forced_step(a, cfg, act, f"periodic packed lattice {t}")
        b.step(torch.from_numpy(act).cuda())
        _same(torch, a, b, f"periodic packed {t}")


@pytest.mark.parametrize("n_a,kw,on", [(30, dict(d_sen=0.3), True), (30, dict(d_sen=0.42, r_avoid=0.2), True),
                                       (20, {}, True), (16, {}, True), (30, dict(d_sen=0.46), False), (10, {}, False),
                                       (30, dict(num_obs_grid_max=30, num_occupied_grid_max=40), True)])
def test_lattice_path_other_radii(torch_cuda, n_a, kw, on):
    """Sensing and avoidance radii other than the defaults: the lattice window (15 x 15 sites for d_sen, 7 x 7 for
    r_avoid / 2) covers d_sen <= 7.4 l_cell and r_avoid / 2 <= 3.4 l_cell; beyond that the call falls back to testing
    every cell by itself.  Either way: equal to the all-cells path and to the oracle."""
    torch = torch_cuda
    from gpu_util import cfg_from_env, teacher_forced_step
    E = 40
    a, b = _pair(E, n_a=n_a, seed=31, **kw)
    a.reset(); b.reset()
    assert a.step_schedule() == (4 + 16 if on else 4), (a.r_avoid, a.params.d_sen)
    cfg = cfg_from_env(a)
    rng = np.random.default_rng(5)
    grid = a.grid_center.cpu().numpy(); n_g = a.n_g.cpu().numpy()
    p = a.p.cpu().numpy()
    for e in range(0, E, 2):                                         # every other environment packed into its shape
        p[e] = grid[e][:, rng.choice(n_g[e], n_a, replace=False)] + rng.normal(0, 0.03, (2, n_a))
    for env in (a, b):
        env.p.copy_(torch.from_numpy(p)); env.observe()
    _same(torch, a, b, "observe")
    for t in range(6):
        act = rng.uniform(-1, 1, (E, 2, n_a)).astype(np.float32)
        teacher_forced_step(a, cfg, act, f"radii {kw} step {t}")
        b.step(torch.from_numpy(act).cuda())
        _same(torch, a, b, f"step {t}")
