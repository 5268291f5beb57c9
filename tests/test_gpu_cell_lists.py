"""Large swarms (256 agents and more): the contact and nearest-neighbour passes through cell lists against the same
passes over all agents (asm_set_step_schedule(+16)) -- bit-equal in every output -- and against the oracle: spread
out, clustered, piled up in one spot (more screened contacts than the per-agent list holds), outside the walls."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
NAMES = ("obs", "reward", "a_prior", "p", "dp", "neighbor_index", "in_flags", "nearest_cell", "sensed_index", "occupied_index")


@pytest.fixture
def lists_off():
    """Call to run the following launches without cell lists; restored afterwards."""
    from marl_amd import _lib
    lib = _lib.load()
    yield lambda off: lib.asm_set_step_schedule(4 + int(off) if off else -1)
    lib.asm_set_step_schedule(-1)


def _both(torch, env, lists_off, fn, tag, off=16):
    """fn(env) from the same state under schedule 4 + off (16: no cell lists; 32: the three-launch step) and under the
    default schedule; returns with the default schedule's results in place."""
    state = {k: getattr(env, k).clone() for k in ("p", "dp", "neighbor_index", "in_flags", "nearest_cell", "stats")}
    lists_off(off)
    assert env.step_schedule() & (64 + 128) == (0 if off == 16 else 64 + 128)
    fn(env)
    want = {k: getattr(env, k).clone() for k in NAMES + ("stats",)}
    for k, v in state.items():
        getattr(env, k).copy_(v)
    lists_off(False)
    assert env.step_schedule() & (64 + 128) == 64
    fn(env)
    for k in NAMES:
        assert torch.equal(getattr(env, k), want[k]), f"{tag}: {k}"
    assert torch.equal(env.stats, want["stats"]), f"{tag}: stats"


@pytest.mark.parametrize("n_a,E,lattice", [(256, 6, True), (300, 5, False), (1024, 3, True), (1024, 2, False)])
def test_cell_lists_equal_all_agent_passes_and_oracle(torch_cuda, lists_off, n_a, E, lattice):
    torch = torch_cuda
    from gpu_util import cfg_from_env, teacher_forced_step
    from marl_amd.batched import BatchedAssemblyEnv
    env = BatchedAssemblyEnv(E, n_a, seed=n_a + E, keep_indices=True, use_lattice=lattice)
    env.reset()
    assert env.step_schedule() & 64 == 64
    for t in range(6):                                                    # half of the resets start clustered (ENV:197-205)
        act = env.random_actions(step=t)
        _both(torch, env, lists_off, lambda e: e.step(act), f"random {t}")
    cfg = cfg_from_env(env)
    rng = np.random.default_rng(n_a)
    for t in range(2):
        act = rng.uniform(-1, 1, (E, 2, n_a)).astype(np.float32)
        teacher_forced_step(env, cfg, act, f"oracle {t}")


def test_cell_lists_adversarial_states(torch_cuda, lists_off):
    """One environment each: everybody inside a 3 cm box (dozens of screened contacts per agent: the list overflows and
    the agent falls back to the pass over all agents); a 30 cm box; agents outside the walls and far away; two tight
    clumps in opposite corners; agents exactly on bin borders."""
    torch = torch_cuda
    from gpu_util import cfg_from_env, check_observation, teacher_forced_step
    from marl_amd.batched import BatchedAssemblyEnv
    n_a, E = 512, 6
    env = BatchedAssemblyEnv(E, n_a, seed=77, keep_indices=True)
    env.reset()
    rng = np.random.default_rng(5)
    p = env.p.cpu().numpy().astype(np.float64)
    p[0] = rng.uniform(-0.015, 0.015, (2, n_a)) + [[1.0], [-0.5]]
    p[1] = rng.uniform(-0.15, 0.15, (2, n_a)) + [[-2.3], [2.3]]
    p[2] = rng.uniform(-2.4, 2.4, (2, n_a)); p[2][:, :40] = rng.uniform(2.4, 2.6, (2, 40)); p[2][:, 40:60] = rng.uniform(-50, 50, (2, 20))
    p[2][:, 60] = [1e6, -1e6]
    p[3][:, :256] = rng.uniform(-0.05, 0.05, (2, 256)) + 2.35; p[3][:, 256:] = rng.uniform(-0.05, 0.05, (2, 256)) - 2.35
    h = 0.07 * 1.001                                                    # contact bins: multiples of the bin side from the corner
    p[4] = -2.4 + h * rng.integers(0, 68, (2, n_a)) + rng.choice([0.0, 1e-7, -1e-7], (2, n_a))
    pf = p.astype(np.float32)
    for e in range(E):                                                   # coincident agents are outside the contract
        _, first = np.unique(pf[e].T, axis=0, return_index=True)
        dup = np.setdiff1d(np.arange(n_a), first)
        pf[e][:, dup] += rng.uniform(1e-4, 1e-3, (2, len(dup))).astype(np.float32)
    env.p.copy_(torch.from_numpy(pf)); env.dp.mul_(0.1)
    _both(torch, env, lists_off, lambda e: e.observe(want_reward=True), "observe")
    cfg = cfg_from_env(env)
    check_observation(env, cfg, True, "adversarial observe vs oracle")
    for t in range(3):
        act = rng.uniform(-1, 1, (E, 2, n_a)).astype(np.float32)
        a = torch.from_numpy(act).cuda()
        _both(torch, env, lists_off, lambda e: e.step(a), f"adversarial step {t}")
    teacher_forced_step(env, cfg, rng.uniform(-1, 1, (E, 2, n_a)).astype(np.float32), "adversarial vs oracle")


def test_cell_lists_other_geometry(torch_cuda, lists_off):
    """A sensing radius that is large against the arena, a small arena, thick agents: the bin geometry follows."""
    torch = torch_cuda
    from gpu_util import cfg_from_env, teacher_forced_step
    from marl_amd.batched import BatchedAssemblyEnv
    rng = np.random.default_rng(9)
    for ctor, over in ((dict(d_sen=1.3), {}), (dict(d_sen=0.15), dict(size_a=0.06)), ({}, dict(half_w=1.6, half_h=3.0))):
        env = BatchedAssemblyEnv(3, 400, seed=21, keep_indices=True, **ctor)
        cfg = cfg_from_env(env)
        for k, v in over.items():
            setattr(env.params, k, v); setattr(cfg, k, v)
        env.reset()
        assert env.step_schedule() & 64 == 64
        for t in range(3):
            act = env.random_actions(step=t)
            _both(torch, env, lists_off, lambda e: e.step(act), f"{ctor} {over} step {t}")
        teacher_forced_step(env, cfg, rng.uniform(-1, 1, (3, 2, 400)).astype(np.float32), f"{ctor} {over} vs oracle")


@pytest.mark.parametrize("n_a,E", [(512, 5), (700, 3), (1024, 3)])
def test_split_step_equals_one_workgroup_per_environment(torch_cuda, lists_off, n_a, E):
    """512 agents and more on the lattice path: the step as three launches with the observation split over 256-agent
    workgroups (asm_set_step_schedule(+32)) against the single launch, every output and the statistics bit for bit,
    random and packed into the shapes; then against the oracle."""
    torch = torch_cuda
    from gpu_util import cfg_from_env, teacher_forced_step
    from marl_amd.batched import BatchedAssemblyEnv
    env = BatchedAssemblyEnv(E, n_a, seed=n_a, keep_indices=True)
    env.reset()
    assert env.step_schedule() & (32 + 64 + 128) == 32 + 64
    for t in range(5):
        act = env.random_actions(step=t)
        _both(torch, env, lists_off, lambda e: e.step(act), f"random {t}", off=32)
    g = torch.Generator(device="cuda").manual_seed(4)
    cell = (torch.rand(E, n_a, device="cuda", generator=g) * env.n_g[:, None]).long()
    pos = torch.gather(env.grid_center, 2, cell[:, None, :].expand(E, 2, n_a)) + 0.03 * torch.randn(E, 2, n_a, device="cuda", generator=g)
    env.p.copy_(pos); env.observe()
    cfg = cfg_from_env(env)
    rng = np.random.default_rng(n_a)
    for t in range(3):
        act = torch.from_numpy(rng.uniform(-1, 1, (E, 2, n_a)).astype(np.float32)).cuda()
        _both(torch, env, lists_off, lambda e: e.step(act), f"packed {t}", off=32)
    assert int((env.sensed_index >= 0).sum(-1).max()) == 80
    lists_off(32)
    teacher_forced_step(env, cfg, rng.uniform(-1, 1, (E, 2, n_a)).astype(np.float32), "split vs oracle")
    env.invalidate_lattice()                                              # no lattice description: back to the single launch
    assert env.step_schedule() & 128 == 0
    teacher_forced_step(env, cfg, rng.uniform(-1, 1, (E, 2, n_a)).astype(np.float32), "after invalidate vs oracle")


@pytest.mark.parametrize("ctor", [dict(num_obs_grid_max=40), dict(num_obs_grid_max=16, d_sen=0.3), dict(is_con_self_state=False),
                                  dict(training_method="rl"), dict(keep_stats=False)])
def test_large_swarm_kernel_other_parameters(torch_cuda, ctor):
    """The large swarm's own instantiation of the step kernel with other list capacities (most lists sub-sampled, so the
    reduction of a list to the cells it keeps and the wave-wide reward sums carry the step), without the self-state
    rows, without the prior, without statistics: against the oracle, from the reference's reset and with every agent
    packed into its shape (full lists, the occupied-cell filter, rewards)."""
    torch = torch_cuda
    from gpu_util import cfg_from_env, teacher_forced_step
    from marl_amd.batched import BatchedAssemblyEnv
    E, n_a = 3, 512
    env = BatchedAssemblyEnv(E, n_a, seed=31, keep_indices=True, **ctor)
    cfg = cfg_from_env(env)
    env.reset()
    assert env.step_schedule() == 32 + 64 + 256
    rng = np.random.default_rng(3)
    for t in range(2):
        teacher_forced_step(env, cfg, rng.uniform(-1, 1, (E, 2, n_a)).astype(np.float32), f"{ctor} step {t}")
    g = torch.Generator(device="cuda").manual_seed(4)
    cell = (torch.rand(E, n_a, device="cuda", generator=g) * env.n_g[:, None]).long()
    pos = torch.gather(env.grid_center, 2, cell[:, None, :].expand(E, 2, n_a))
    env.p.copy_(pos + 0.03 * torch.randn(E, 2, n_a, device="cuda", generator=g))
    env.observe()
    for t in range(3):
        teacher_forced_step(env, cfg, rng.uniform(-1, 1, (E, 2, n_a)).astype(np.float32), f"{ctor} packed step {t}")
    if ctor.get("num_obs_grid_max"):
        assert int((env.sensed_index[:, :, -1] >= 0).sum()) > 0        # lists that fill every slot: sub-sampled or full
