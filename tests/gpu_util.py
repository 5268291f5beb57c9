"""Shared helpers of the GPU parity tests: batched oracle evaluation (CPU, double)
next to the HIP path (GPU, fp32 state) on the same numbers."""
import numpy as np

import oracle_lib as ol

RTOL = 1e-5      # the contract: values within 1e-5 relative of the reference path (fp32 state)
ATOL = 1e-7      # absolute floor for values that are exactly 0 in one of the two


def ulp_diff_f32(a, b):
    """Distance in float32 units-in-the-last-place between a (float32) and b (float64
    reference rounded to float32)."""
    a = np.ascontiguousarray(a, dtype=np.float32); b = np.ascontiguousarray(b).astype(np.float32)
    ai = a.view(np.int32).astype(np.int64); bi = b.view(np.int32).astype(np.int64)
    ai = np.where(ai < 0, -(ai & 0x7fffffff), ai); bi = np.where(bi < 0, -(bi & 0x7fffffff), bi)
    return np.abs(ai - bi)


def assert_close(got_f32, want_f64, what, max_ulp=1):
    got = np.asarray(got_f32); want = np.asarray(want_f64)
    np.testing.assert_allclose(got.astype(np.float64), want, rtol=RTOL, atol=ATOL, err_msg=what)
    # tighter, informative bound: one rounding to float32 of the double-precision result
    u = ulp_diff_f32(got, want)
    assert u.max() <= max_ulp, f"{what}: {u.max()} float32 ulps off (allowed {max_ulp})"


def cfg_from_env(env):
    return ol.make_cfg(n_a=env.n_a, r_avoid=env.r_avoid, is_boundary=bool(env.params.is_boundary),
                       is_con_self_state=bool(env.params.is_con_self_state), d_sen=env.params.d_sen,
                       llm_rl=env.training_method == "llm_rl", num_obs_grid_max=env.params.num_obs_grid_max,
                       num_occupied_grid_max=env.params.num_occupied_grid_max)


def oracle_observe_batch(cfg, p, dp, grid, n_g, l_cell):
    """p, dp: (E,2,n) float; grid (E,2,n_g_max). Returns dict of stacked oracle outputs."""
    E = p.shape[0]
    out = {k: [] for k in ("obs", "rew", "nbr", "in_flags", "sensed", "occupied")}
    for e in range(E):
        g = np.ascontiguousarray(grid[e][:, :n_g[e]], dtype=np.float64)
        o, r, nb, inf, sen, occ = ol.env_observe(cfg, np.ascontiguousarray(p[e], dtype=np.float64),
                                                 np.ascontiguousarray(dp[e], dtype=np.float64), g, float(l_cell[e]))
        for k, v in zip(out, (o, r, nb, inf, sen, occ)):
            out[k].append(v)
    return {k: np.stack(v) for k, v in out.items()}


def oracle_dynamics_batch(cfg, p, dp, act, grid, n_g, l_cell, nbr):
    E = p.shape[0]
    P = np.array(p, dtype=np.float64); V = np.array(dp, dtype=np.float64)
    prior = np.zeros_like(P); contacts = np.zeros(E, np.int64)
    for e in range(E):
        g = np.ascontiguousarray(grid[e][:, :n_g[e]], dtype=np.float64)
        pe = np.ascontiguousarray(P[e]); ve = np.ascontiguousarray(V[e])
        ap, co = ol.env_dynamics(cfg, pe, ve, np.ascontiguousarray(act[e], dtype=np.float32), g, float(l_cell[e]),
                                 np.ascontiguousarray(nbr[e], dtype=np.int32))
        P[e] = pe; V[e] = ve; prior[e] = ap; contacts[e] = np.triu(co, 1).sum()
    return P, V, prior, contacts


def check_observation(env, cfg, want_reward, tag=""):
    """Stage B: oracle fed the device's own fp32 state; indices exact, values 1e-5."""
    p = env.p.cpu().numpy(); dp = env.dp.cpu().numpy(); grid = env.grid_center.cpu().numpy()
    n_g = env.n_g.cpu().numpy(); l_cell = env.l_cell.cpu().numpy()
    o = oracle_observe_batch(cfg, p, dp, grid, n_g, l_cell)
    assert np.array_equal(env.neighbor_index.cpu().numpy(), o["nbr"]), f"{tag} neighbor_index"
    assert np.array_equal(env.in_flags.cpu().numpy(), o["in_flags"]), f"{tag} in_flags"
    if env.sensed_index is not None:
        assert np.array_equal(env.sensed_index.cpu().numpy(), o["sensed"]), f"{tag} sensed_index"
        assert np.array_equal(env.occupied_index.cpu().numpy(), o["occupied"]), f"{tag} occupied_index"
    assert_close(env.obs.cpu().numpy(), o["obs"], f"{tag} obs")
    if want_reward:
        assert np.array_equal(env.reward.cpu().numpy().astype(np.float64), o["rew"]), f"{tag} reward"
    return o


def teacher_forced_step(env, cfg, act, tag=""):
    """One device step checked in two stages against the oracle.
    Stage A: same (p, dp, a, previous neighbour list) -> p', dp', a_prior.
    Stage B: oracle fed the device's p', dp' -> obs, reward, indices."""
    import torch
    p0 = env.p.cpu().numpy(); dp0 = env.dp.cpu().numpy(); nbr0 = env.neighbor_index.cpu().numpy()
    grid = env.grid_center.cpu().numpy(); n_g = env.n_g.cpu().numpy(); l_cell = env.l_cell.cpu().numpy()
    P, V, prior, contacts = oracle_dynamics_batch(cfg, p0, dp0, act, grid, n_g, l_cell, nbr0)
    obs, rew, done, info, a_prior = env.step(torch.from_numpy(np.ascontiguousarray(act)).to(env.device))
    assert_close(env.p.cpu().numpy(), P, f"{tag} p")
    assert_close(env.dp.cpu().numpy(), V, f"{tag} dp")
    if a_prior is not None:
        assert_close(a_prior.cpu().numpy(), prior, f"{tag} a_prior")
    assert not done.any()
    o = check_observation(env, cfg, True, tag)
    return o, contacts
