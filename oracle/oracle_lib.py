"""TEST INFRASTRUCTURE — ctypes binding of oracle/libasm_oracle.so (numpy in/out).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libasm_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libAssemblyEnv.so")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_bp = C.POINTER(C.c_bool)
_fp = C.POINTER(C.c_float)


class OracleCfg(C.Structure):
    _fields_ = [("n_a", C.c_int), ("topo_nei_max", C.c_int), ("num_obs_grid_max", C.c_int),
                ("num_occupied_grid_max", C.c_int), ("is_boundary", C.c_int),
                ("is_con_self_state", C.c_int), ("llm_rl", C.c_int), ("size_a", C.c_double),
                ("d_sen", C.c_double), ("r_avoid", C.c_double), ("Vel_max", C.c_double),
                ("k_ball", C.c_double), ("k_wall", C.c_double), ("c_wall", C.c_double),
                ("dt", C.c_double), ("sensitivity", C.c_double), ("mass", C.c_double),
                ("half_w", C.c_double), ("half_h", C.c_double)]


def build(force: bool = False) -> None:
    if force or not os.path.isfile(SO) or os.path.getmtime(SO) < os.path.getmtime(os.path.join(HERE, "asm_oracle.c")):
        subprocess.check_call(["make", "-C", HERE, "libasm_oracle.so"], stdout=subprocess.DEVNULL)


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(SO)
        _lib.oracle_coverage_rate.restype = C.c_double
        _lib.oracle_distribution_uniformity.restype = C.c_double
        _lib.oracle_voronoi_uniformity.restype = C.c_double
        _lib.oracle_avg_distance_to_target.restype = C.c_double
        _lib.oracle_count_collisions.restype = C.c_int
        _lib.oracle_run_envs.restype = C.c_double
        _lib.oracle_obs_dim.restype = C.c_int
        _lib.oracle_bind.restype = C.c_int
        _lib.oracle_bind.argtypes = [C.c_char_p]
    return _lib


def d(a):
    assert a.dtype == np.float64 and a.flags.c_contiguous
    return a.ctypes.data_as(_dp)


def i32(a):
    assert a.dtype == np.int32 and a.flags.c_contiguous
    return a.ctypes.data_as(_ip)


def b8(a):
    assert a.dtype == np.bool_ and a.flags.c_contiguous
    return a.ctypes.data_as(_bp)


def f32(a):
    assert a.dtype == np.float32 and a.flags.c_contiguous
    return a.ctypes.data_as(_fp)


def r_avoid_for(n_a: int, n_gs, l_cells) -> float:
    """ENV:121-122."""
    return round(float(np.sqrt(4 * np.min(n_gs) / (n_a * np.pi)) * np.min(l_cells)), 2)


def make_cfg(n_a=30, r_avoid=0.26, is_boundary=True, is_con_self_state=True, llm_rl=True, d_sen=0.4,
             topo_nei_max=6, num_obs_grid_max=80, num_occupied_grid_max=200) -> OracleCfg:
    c = OracleCfg()
    lib().oracle_default_cfg(C.byref(c), C.c_int(n_a), C.c_double(r_avoid))
    c.is_boundary = int(is_boundary)
    c.is_con_self_state = int(is_con_self_state)
    c.llm_rl = int(llm_rl)
    c.d_sen = d_sen
    c.topo_nei_max = topo_nei_max
    c.num_obs_grid_max = num_obs_grid_max
    c.num_occupied_grid_max = num_occupied_grid_max
    return c


def obs_dim(cfg: OracleCfg) -> int:
    return lib().oracle_obs_dim(C.byref(cfg))


def bind(so_path=None) -> None:
    rc = lib().oracle_bind(None if so_path is None else so_path.encode())
    if rc != 0:
        raise RuntimeError(f"oracle_bind({so_path}) failed: {rc}")


class FiveFns:
    """The five reference entry points, callable on numpy arrays, from either the
    oracle restatement (prefix 'oracle_') or a library exporting the reference names."""

    def __init__(self, cdll: C.CDLL, prefixed: bool):
        g = (lambda n: getattr(cdll, ("oracle" + n) if prefixed else n))
        self.f_obs = g("_get_observation")
        self.f_rew = g("_get_reward")
        self.f_sf = g("_sf_b2b_all")
        self.f_b2w = g("_get_dist_b2w")
        self.f_prior = getattr(cdll, "oracle_calculateActionPrior" if prefixed else "calculateActionPrior")

    def get_observation(self, p, dp, heading, boundary_pos, grid, d_sen, r_avoid, l_cell, Vel_max=0.8,
                        topo=6, n_obs=80, n_occ=200, cond=(False, True, True, False), obs_dim=None):
        n_a = p.shape[1]; n_g = grid.shape[1]
        if obs_dim is None:
            obs_dim = 2 * 2 * (topo + 1 + (1 if cond[2] else 0)) + 2 * n_obs if cond[1] else 2 * 2 * (topo + (1 if cond[2] else 0)) + 6
        obs = np.zeros((obs_dim, n_a)); nbr = -np.ones((n_a, topo), np.int32)
        inf = np.zeros(n_a, np.int32); sen = -np.ones((n_a, n_obs), np.int32); occ = -np.ones((n_a, n_occ), np.int32)
        c = np.array(cond, dtype=np.bool_)
        self.f_obs(d(p), d(dp), d(heading), d(obs), d(boundary_pos), d(grid), i32(nbr), i32(inf), i32(sen), i32(occ),
                   C.c_double(d_sen), C.c_double(r_avoid), C.c_double(l_cell), C.c_double(Vel_max), C.c_int(topo),
                   C.c_int(n_obs), C.c_int(n_occ), C.c_int(n_a), C.c_int(n_g), C.c_int(obs_dim), C.c_int(2), b8(c))
        return obs, nbr, inf, sen, occ

    def get_reward(self, p, dp, heading, act, boundary_pos, grid, nbr, inf, sen, occ, d_sen, r_avoid, l_cell,
                   cond=(False, True, True, True, True)):
        n_a = p.shape[1]; n_g = grid.shape[1]
        rew = np.zeros((1, n_a)); c = np.array(cond, dtype=np.bool_)
        cb = np.zeros((n_a, n_a), np.bool_); cw = np.zeros((4, n_a), np.bool_); coef = np.array([0.05])
        self.f_rew(d(p), d(dp), d(heading), d(np.ascontiguousarray(act, dtype=np.float64)), d(rew), d(boundary_pos),
                   d(grid), i32(nbr), i32(inf), i32(sen), i32(occ), C.c_double(d_sen), C.c_double(r_avoid),
                   C.c_double(l_cell), C.c_int(nbr.shape[1]), C.c_int(sen.shape[1]), C.c_int(occ.shape[1]),
                   C.c_int(n_a), C.c_int(n_g), C.c_int(2), b8(c), b8(cb), b8(cw), d(coef))
        return rew

    def sf_b2b_all(self, p, d_edge, collide, boundary_pos, d_center, k_ball=30.0, periodic=False):
        n_a = p.shape[1]; sf = np.zeros((2, n_a))
        self.f_sf(d(p), d(sf), d(d_edge), b8(collide), d(boundary_pos), d(d_center), C.c_int(n_a), C.c_int(2),
                  C.c_double(k_ball), C.c_bool(periodic))
        return sf

    def get_dist_b2w(self, p, size, boundary_pos):
        n_a = p.shape[1]; dw = np.ones((4, n_a)); cw = np.zeros((4, n_a), np.bool_)
        self.f_b2w(d(p), d(size), d(dw), b8(cw), C.c_int(2), C.c_int(n_a), d(boundary_pos))
        return dw, cw

    def prior(self, p, dp, grid, nbr, d_sen, r_avoid, l_cell):
        n_a = p.shape[1]; n_g = grid.shape[1]; ap = np.zeros((2, n_a))
        self.f_prior(d(p), d(dp), d(ap), d(grid), i32(nbr), C.c_double(d_sen), C.c_double(r_avoid), C.c_double(l_cell),
                     C.c_int(nbr.shape[1]), C.c_int(n_a), C.c_int(n_g), C.c_int(2))
        return ap


def oracle_fns() -> FiveFns:
    return FiveFns(lib(), True)


def reference_fns() -> FiveFns:
    """The compiled reference (oracle/_ref); only where it has been built."""
    return FiveFns(C.CDLL(REF_SO), False)


def get_dist_b2b(p, size_a=0.035, periodic=False, half_w=2.4, half_h=2.4):
    n_a = p.shape[1]
    dc = np.zeros((n_a, n_a)); de = np.zeros((n_a, n_a)); co = np.zeros((n_a, n_a), np.bool_)
    lib().oracle_get_dist_b2b(d(p), C.c_int(n_a), C.c_double(size_a), C.c_bool(periodic), C.c_double(half_w),
                              C.c_double(half_h), d(dc), d(de), b8(co))
    return dc, de, co


def env_dynamics(cfg, p, dp, act, grid, l_cell, nbr):
    """In-place on p, dp.  Returns (a_prior, is_collide_b2b)."""
    n = cfg.n_a; ap = np.zeros((2, n)); co = np.zeros((n, n), np.bool_)
    lib().oracle_env_dynamics(C.byref(cfg), d(p), d(dp), f32(act), d(grid), C.c_int(grid.shape[1]),
                              C.c_double(l_cell), i32(nbr), d(ap), b8(co))
    return ap, co


def env_observe(cfg, p, dp, grid, l_cell, want_reward=True):
    n = cfg.n_a; od = obs_dim(cfg)
    obs = np.zeros((od, n)); nbr = -np.ones((n, cfg.topo_nei_max), np.int32); inf = np.zeros(n, np.int32)
    sen = -np.ones((n, cfg.num_obs_grid_max), np.int32); occ = -np.ones((n, cfg.num_occupied_grid_max), np.int32)
    rew = np.zeros((1, n))
    lib().oracle_env_observe(C.byref(cfg), d(p), d(dp), d(grid), C.c_int(grid.shape[1]), C.c_double(l_cell),
                             i32(nbr), d(obs), d(rew) if want_reward else None, i32(inf), i32(sen), i32(occ))
    return obs, rew, nbr, inf, sen, occ


def env_step(cfg, p, dp, act, grid, l_cell, nbr):
    """In-place on p, dp, nbr.  Returns (obs, rew, a_prior, in_flags, sensed, occupied, collide)."""
    n = cfg.n_a; od = obs_dim(cfg)
    obs = np.zeros((od, n)); rew = np.zeros((1, n)); ap = np.zeros((2, n)); inf = np.zeros(n, np.int32)
    sen = -np.ones((n, cfg.num_obs_grid_max), np.int32); occ = -np.ones((n, cfg.num_occupied_grid_max), np.int32)
    co = np.zeros((n, n), np.bool_)
    lib().oracle_env_step(C.byref(cfg), d(p), d(dp), f32(act), d(grid), C.c_int(grid.shape[1]), C.c_double(l_cell),
                          i32(nbr), d(obs), d(rew), d(ap), i32(inf), i32(sen), i32(occ), b8(co))
    return obs, rew, ap, inf, sen, occ, co


def rule_actions(cfg, p, dp, grid, l_cell):
    """ENV:525-596 on the given state -> (2, n_a) float64."""
    act = np.zeros((2, cfg.n_a))
    lib().oracle_rule_actions(C.byref(cfg), d(p), d(dp), d(grid), C.c_int(grid.shape[1]), C.c_double(l_cell), d(act))
    return act


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr); k = (C.c_uint32 * 2)(*key); o = (C.c_uint32 * 4)()
    lib().oracle_philox4x32(c, k, o)
    return list(o)


def pack_shapes(grids, l_cells):
    """Concatenate shape grids (n_g_k, 2) -> (flat, off, ng, lcell) as oracle_batch_reset expects."""
    off = np.zeros(len(grids), np.int32); ng = np.array([g.shape[0] for g in grids], np.int32)
    off[1:] = np.cumsum(ng)[:-1]
    flat = np.ascontiguousarray(np.concatenate(grids, axis=0), dtype=np.float64)
    return flat, off, ng, np.array(l_cells, dtype=np.float64)


def batch_reset(seed, episode, env0, E, n_a, grids, l_cells, n_g_max, half_w=2.4, half_h=2.4):
    flat, off, ng, lc = pack_shapes(grids, l_cells)
    sid = np.zeros(E, np.int32); p = np.zeros((E, 2, n_a)); dp = np.zeros((E, 2, n_a))
    grid = np.zeros((E, 2, n_g_max)); n_g = np.zeros(E, np.int32); l_cell = np.zeros(E)
    lib().oracle_batch_reset(C.c_uint64(seed), C.c_uint32(episode), C.c_int64(env0), C.c_int(E), C.c_int(n_a),
                             C.c_int(len(grids)), d(flat), i32(off), i32(ng), d(lc), C.c_int(n_g_max),
                             C.c_double(half_w), C.c_double(half_h), i32(sid), d(p), d(dp), d(grid), i32(n_g), d(l_cell))
    return sid, p, dp, grid, n_g, l_cell


def run_envs(cfg, steps, threads, p, dp, act, grid, n_g, l_cell):
    E = p.shape[0]
    return lib().oracle_run_envs(C.byref(cfg), C.c_int(E), C.c_int(steps), C.c_int(threads), d(p), d(dp), f32(act),
                                 d(grid), C.c_int(grid.shape[2]), i32(n_g), d(l_cell))


def metrics(p, grid, r_avoid, collide=None):
    n_a = p.shape[1]; n_g = grid.shape[1]; L = lib()
    out = {
        "coverage_rate": L.oracle_coverage_rate(d(p), C.c_int(n_a), d(grid), C.c_int(n_g), C.c_double(r_avoid)),
        "distribution_uniformity": L.oracle_distribution_uniformity(d(p), C.c_int(n_a)),
        "voronoi_uniformity": L.oracle_voronoi_uniformity(d(p), C.c_int(n_a), d(grid), C.c_int(n_g)),
        "avg_distance_to_target": L.oracle_avg_distance_to_target(d(p), C.c_int(n_a), d(grid), C.c_int(n_g)),
    }
    if collide is not None:
        out["num_collisions"] = L.oracle_count_collisions(b8(collide), C.c_int(n_a))
    return out
