"""The five reference-named entry points of libAssemblyEnv.so (host double pointers,
computed on the GPU in double) against the CPU oracle and the golden vectors:
bit-exact on every index and on every value that involves no transcendental."""
import glob
import os

import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROLLOUTS = sorted(glob.glob(os.path.join(GOLDEN, "rollout_*.npz")))
BPOS = np.array([-2.4, 2.4, 2.4, -2.4])


@pytest.fixture(scope="module")
def hip():
    from marl_amd import _lib
    lib = _lib.load()
    assert lib.asm_device_count() >= 1, lib.asm_last_error()
    return ol.FiveFns(lib, False)


@pytest.mark.parametrize("path", ROLLOUTS, ids=[os.path.basename(p)[8:-4] for p in ROLLOUTS])
def test_five_symbols_on_golden_states(hip, path):
    z = np.load(path)
    orc = ol.oracle_fns()
    grid = np.ascontiguousarray(z["grid"]); l_cell = float(z["l_cell"]); d_sen = float(z["d_sen"]); r_avoid = float(z["r_avoid"])
    periodic = not bool(z["is_boundary"]); con_self = bool(z["is_con_self_state"])
    T = z["act"].shape[0]
    for t in range(0, T, max(1, T // 4)):
        p = np.ascontiguousarray(z["p"][t]); dp = np.ascontiguousarray(z["dp"][t]); heading = np.zeros_like(p)
        cond = (periodic, True, con_self, False)
        obs, nbr, inf, sen, occ = hip.get_observation(p, dp, heading, BPOS, grid, d_sen, r_avoid, l_cell, cond=cond)
        # straight against what the reference produced for this state
        assert np.array_equal(obs, z["obs"][t]) and np.array_equal(nbr, z["nbr"][t])
        assert np.array_equal(inf, z["in_flags"][t]) and np.array_equal(sen, z["sensed"][t])
        assert np.array_equal(occ, z["occupied"][t])
        rew = hip.get_reward(p, dp, heading, dp, BPOS, grid, nbr, inf, sen, occ, d_sen, r_avoid, l_cell,
                             cond=(periodic, True, True, True, True))
        assert np.array_equal(rew, z["rew"][t])
        # dynamics pieces on the state BEFORE step t
        p0 = np.ascontiguousarray(z["p"][t - 1] if t else z["p0"]); dp0 = np.ascontiguousarray(z["dp"][t - 1] if t else z["dp0"])
        nbr0 = np.ascontiguousarray(z["nbr"][t - 1] if t else z["nbr0"])
        dc, de, co = ol.get_dist_b2b(p0, periodic=periodic)
        sf_h = hip.sf_b2b_all(p0, de, co, BPOS, dc, 30.0, periodic)
        assert np.array_equal(sf_h, orc.sf_b2b_all(p0, de, co, BPOS, dc, 30.0, periodic))
        size = np.full(p0.shape[1], 0.035)
        for a, b in zip(hip.get_dist_b2w(p0, size, BPOS), orc.get_dist_b2w(p0, size, BPOS)):
            assert np.array_equal(a, b)
        pr = hip.prior(p0, dp0, grid, nbr0, d_sen, r_avoid, l_cell)
        assert np.array_equal(pr, z["prior"][t])


def test_non_cartesian_and_big_radius(hip):
    orc = ol.oracle_fns()
    z = np.load(os.path.join(GOLDEN, "fn_occ200.npz"))
    p = np.ascontiguousarray(z["p"]); dp = np.ascontiguousarray(z["dp"]); grid = np.ascontiguousarray(z["grid"])
    heading = np.random.default_rng(0).uniform(-1, 1, p.shape)
    d_sen, r_avoid, l_cell = float(z["d_sen"]), float(z["r_avoid"]), float(z["l_cell"])
    got = hip.get_observation(p, dp, heading, BPOS, grid, d_sen, r_avoid, l_cell)
    for name, g in zip(("obs", "nbr", "in_flags", "sensed", "occupied"), got):
        assert np.array_equal(g, z[name]), name
    for cond in [(False, False, True, False), (True, False, False, False), (True, True, True, False)]:
        a = hip.get_observation(p, dp, heading, BPOS, grid, 0.4, 0.26, l_cell, cond=cond)
        b = orc.get_observation(p, dp, heading, BPOS, grid, 0.4, 0.26, l_cell, cond=cond)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)


def test_walls_and_contacts_crowded_corner(hip):
    """60 agents pushed into a wall corner: many contacts and wall hits."""
    orc = ol.oracle_fns()
    rng = np.random.default_rng(5)
    p = np.ascontiguousarray(np.stack([rng.uniform(2.0, 2.45, 60), rng.uniform(-2.45, -2.0, 60)]))
    dc, de, co = ol.get_dist_b2b(p)
    assert co.sum() > 20
    assert np.array_equal(hip.sf_b2b_all(p, de, co, BPOS, dc), orc.sf_b2b_all(p, de, co, BPOS, dc))
    size = np.full(60, 0.035)
    a = hip.get_dist_b2w(p, size, BPOS); b = orc.get_dist_b2w(p, size, BPOS)
    assert a[1].sum() > 5 and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_five_symbols_on_lattice_state(hip):
    """Agents on a binary lattice (equal distances everywhere), true double inputs: every output of the five
    entry points equals the oracle's, values included (no transcendental on this path except the reward's cos)."""
    from marl_amd.shapes import load_shapes
    orc = ol.oracle_fns()
    l_cells, grids = load_shapes()
    rng = np.random.default_rng(12)
    for shape, n_a, q in ((0, 30, 1 / 16), (3, 30, 1 / 64), (5, 64, 1 / 32)):
        grid = np.ascontiguousarray(np.asarray(grids[shape], dtype=np.float64).T); l_cell = float(l_cells[shape])
        m = int(round(0.75 / q)); side = 2 * m + 1
        pick = rng.choice(side * side, n_a, replace=False)
        p = np.ascontiguousarray(np.stack([(pick % side - m) * q, (pick // side - m) * q]).astype(np.float64))
        dp = np.ascontiguousarray(rng.integers(-2, 3, (2, n_a)) * 0.25); heading = np.zeros_like(p)
        a = hip.get_observation(p, dp, heading, BPOS, grid, 0.4, 0.26, l_cell)
        b = orc.get_observation(p, dp, heading, BPOS, grid, 0.4, 0.26, l_cell)
        for name, x, y in zip(("obs", "nbr", "in_flags", "sensed", "occupied"), a, b):
            assert np.array_equal(x, y), (shape, name)
        obs, nbr, inf, sen, occ = b
        ra = hip.get_reward(p, dp, heading, dp, BPOS, grid, nbr, inf, sen, occ, 0.4, 0.26, l_cell)
        rb = orc.get_reward(p, dp, heading, dp, BPOS, grid, nbr, inf, sen, occ, 0.4, 0.26, l_cell)
        assert np.array_equal(ra, rb), shape
        dc, de, co = ol.get_dist_b2b(p)
        assert np.array_equal(hip.sf_b2b_all(p, de, co, BPOS, dc), orc.sf_b2b_all(p, de, co, BPOS, dc))
        assert np.array_equal(hip.prior(p, dp, grid, nbr, 0.4, 0.26, l_cell), orc.prior(p, dp, grid, nbr, 0.4, 0.26, l_cell))


def test_unsupported_argument_does_not_end_the_process(hip):
    """The reference's symbols return void and never end their host (assembly.py:144-145 only prints): a call this
    library cannot run leaves NaN / -1 in the outputs, a message and a status, and the next call works."""
    from marl_amd import _lib
    lib = _lib.load()
    z = np.load(ROLLOUTS[0])
    p = np.ascontiguousarray(z["p"][0]); dp = np.ascontiguousarray(z["dp"][0]); grid = np.ascontiguousarray(z["grid"])
    heading = np.zeros_like(p)
    d_sen, r_avoid, l_cell = float(z["d_sen"]), float(z["r_avoid"]), float(z["l_cell"])
    obs, nbr, inf, sen, occ = hip.get_observation(p, dp, heading, BPOS, grid, d_sen, r_avoid, l_cell, topo=5)
    assert lib.asm_compat_status() != 0 and b"topo_nei_max" in lib.asm_last_error()
    assert np.isnan(obs).all() and (nbr == -1).all() and (inf == -1).all() and (sen == -1).all() and (occ == -1).all()
    with pytest.raises(RuntimeError, match="topo_nei_max"):
        _lib.check_compat("_get_observation")
    pr = hip.prior(p, dp, grid, np.full((p.shape[1], 7), -1, np.int32), d_sen, r_avoid, l_cell)
    assert lib.asm_compat_status() != 0 and np.isnan(pr).all()
    # and the library still works afterwards
    got = hip.get_observation(p, dp, heading, BPOS, grid, d_sen, r_avoid, l_cell,
                              cond=(not bool(z["is_boundary"]), True, bool(z["is_con_self_state"]), False))
    assert lib.asm_compat_status() == 0
    assert np.array_equal(got[0], z["obs"][0]) and np.array_equal(got[1], z["nbr"][0])
