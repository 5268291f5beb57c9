"""The CPU oracle (oracle/asm_oracle.c) against golden vectors recorded from the
reference itself (tests/golden/make_golden.py).  Bit-exact: the oracle is a
double-precision restatement in the reference's operation order."""
import glob
import os

import numpy as np
import pytest

import oracle_lib as ol

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROLLOUTS = sorted(glob.glob(os.path.join(GOLDEN, "rollout_*.npz")))


def _cfg(z):
    return ol.make_cfg(n_a=int(z["n_a"]), r_avoid=float(z["r_avoid"]), is_boundary=bool(z["is_boundary"]),
                       is_con_self_state=bool(z["is_con_self_state"]), d_sen=float(z["d_sen"]))


@pytest.mark.parametrize("path", ROLLOUTS, ids=[os.path.basename(p)[8:-4] for p in ROLLOUTS])
def test_rollout_bit_exact(path):
    z = np.load(path)
    cfg = _cfg(z)
    grid = np.ascontiguousarray(z["grid"]); l_cell = float(z["l_cell"])
    p = z["p0"].copy(); dp = z["dp0"].copy()
    obs0, _, nbr, inf0, sen0, _ = ol.env_observe(cfg, p, dp, grid, l_cell)
    assert np.array_equal(obs0, z["obs0"]) and np.array_equal(nbr, z["nbr0"])
    assert np.array_equal(inf0, z["in_flags0"]) and np.array_equal(sen0, z["sensed0"])
    for t in range(z["act"].shape[0]):
        obs, rew, prior, inf, sen, occ, co = ol.env_step(cfg, p, dp, np.ascontiguousarray(z["act"][t]), grid, l_cell, nbr)
        for name, got in [("obs", obs), ("rew", rew), ("prior", prior), ("p", p), ("dp", dp), ("nbr", nbr),
                          ("in_flags", inf), ("sensed", sen), ("occupied", occ), ("collide", co)]:
            assert np.array_equal(got, z[name][t]), f"{name} differs at step {t}"


def test_teacher_forced_halves_match_full_step():
    """dynamics + observe on a given state == the recorded full step."""
    z = np.load(os.path.join(GOLDEN, "rollout_n30_rule150.npz"))
    cfg = _cfg(z); grid = np.ascontiguousarray(z["grid"]); l_cell = float(z["l_cell"])
    for t in range(1, z["act"].shape[0]):
        p = z["p"][t - 1].copy(); dp = z["dp"][t - 1].copy(); nbr = z["nbr"][t - 1].copy()
        prior, co = ol.env_dynamics(cfg, p, dp, np.ascontiguousarray(z["act"][t]), grid, l_cell, nbr)
        assert np.array_equal(p, z["p"][t]) and np.array_equal(dp, z["dp"][t]) and np.array_equal(prior, z["prior"][t])
        obs, rew, nbr2, inf, sen, occ = ol.env_observe(cfg, p, dp, grid, l_cell)
        assert np.array_equal(obs, z["obs"][t]) and np.array_equal(rew, z["rew"][t])
        assert np.array_equal(sen, z["sensed"][t]) and np.array_equal(occ, z["occupied"][t])


def test_occupied_over_200_and_sensed_over_80():
    z = np.load(os.path.join(GOLDEN, "fn_occ200.npz"))
    f = ol.oracle_fns()
    p = z["p"].copy(); dp = z["dp"].copy(); grid = np.ascontiguousarray(z["grid"])
    heading = np.zeros_like(p); bpos = np.array([-2.4, 2.4, 2.4, -2.4])
    d_sen, r_avoid, l_cell = float(z["d_sen"]), float(z["r_avoid"]), float(z["l_cell"])
    obs, nbr, inf, sen, occ = f.get_observation(p, dp, heading, bpos, grid, d_sen, r_avoid, l_cell)
    assert (z["occupied"] >= 0).sum(1).max() == 200 and (z["sensed"] >= 0).sum(1).max() == 80
    for name, got in [("obs", obs), ("nbr", nbr), ("in_flags", inf), ("sensed", sen), ("occupied", occ)]:
        assert np.array_equal(got, z[name]), name
    rew = f.get_reward(p, dp, heading, np.zeros_like(p), bpos, grid, nbr, inf, sen, occ, d_sen, r_avoid, l_cell)
    assert np.array_equal(rew, z["rew"])
    assert np.array_equal(f.prior(p, dp, grid, nbr, d_sen, r_avoid, l_cell), z["prior"])


def test_metrics_match_wrapper():
    z = np.load(os.path.join(GOLDEN, "metrics.npz"))
    for k in range(int(z["count"])):
        m = ol.metrics(np.ascontiguousarray(z[f"p_{k}"]), np.ascontiguousarray(z[f"grid_{k}"]),
                       float(z[f"r_avoid_{k}"]), np.ascontiguousarray(z[f"collide_{k}"]))
        assert m["coverage_rate"] == float(z[f"coverage_rate_{k}"])
        assert m["num_collisions"] == int(z[f"num_collisions_{k}"])
        # numpy's var/mean use pairwise summation; the oracle sums left to right
        for key in ("distribution_uniformity", "voronoi_uniformity", "avg_distance_to_target"):
            assert m[key] == pytest.approx(float(z[f"{key}_{k}"]), rel=1e-12, abs=1e-14), key


def test_philox_known_answers():
    """Philox4x32-10 known-answer vectors of the Random123 distribution (kat_vectors)."""
    assert ol.philox([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert ol.philox([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert ol.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_batch_reset_ranges_and_determinism():
    from marl_amd.shapes import load_shapes
    l_cells, grids = load_shapes()
    a = ol.batch_reset(226, 0, 0, 64, 30, grids, l_cells, 544)
    b = ol.batch_reset(226, 0, 32, 32, 30, grids, l_cells, 544)     # same envs by global id
    for x, y in zip(a, b):
        assert np.array_equal(x[32:], y)
    sid, p, dp, grid, n_g, l_cell = a
    assert sid.min() >= 0 and sid.max() <= 6 and len(set(sid.tolist())) > 3
    assert np.abs(p).max() <= 2.4 and np.abs(dp).max() <= 0.5
    assert np.array_equal(n_g, np.array([grids[s].shape[0] for s in sid]))
    # grid = R(theta) g + offset: pairwise cell distances are preserved
    g0 = grids[sid[0]]; ge = grid[0][:, :n_g[0]]
    assert np.allclose(np.linalg.norm(ge[:, 1:] - ge[:, :-1], axis=0), np.linalg.norm(g0[1:] - g0[:-1], axis=1), atol=1e-12)
    c = ol.batch_reset(226, 1, 0, 8, 30, grids, l_cells, 544)
    assert not np.array_equal(c[1], p[:8])


def test_oracle_orders_equal_norms_by_index():
    """std::sort (AssemblyEnv.cpp:646) leaves the order of equal norms unspecified; the oracle lists them by
    (squared norm, index)."""
    import oracle_lib as ol
    n_a = 30
    p = np.zeros((2, n_a)); p[0] = np.linspace(-2.3, 2.3, n_a); p[1] = 2.0
    P = np.array([np.float32(-0.74249387), np.float32(0.27645427)], dtype=np.float64)
    d = float(np.float32(0.27645427) - np.float32(0.21534318))
    p[:, 25] = P; p[:, 11] = P + [0, -d]; p[:, 19] = P + [-d, 0]; p[:, 24] = P + [0, -0.01]
    p = np.ascontiguousarray(p.astype(np.float32).astype(np.float64))
    grid = np.ascontiguousarray(np.random.default_rng(0).uniform(-1, 1, (2, 100)))
    o = ol.oracle_fns().get_observation(p, np.zeros((2, n_a)), np.zeros((2, n_a)), np.array([-2.4, 2.4, 2.4, -2.4]),
                                        grid, 0.4, 0.26, 0.06)
    assert o[1][25].tolist() == [24, 11, 19, -1, -1, -1]
