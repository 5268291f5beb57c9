"""Generates the golden vectors under tests/golden/ from the REFERENCE itself.

Run in the build container only (needs /root/reference and oracle/_ref built):

    make -C oracle ref && python tests/golden/make_golden.py

Sources of truth used here:
  * the reference Python environment (cus_gym/.../assembly.py) imported through
    oracle/ref_env.py, driving the compiled reference AssemblyEnv.cpp;
  * the five C entry points of that compiled library called directly.
The fixtures are data only: inputs and the reference's outputs.

Files written (all .npz, compressed):
  rollout_<tag>.npz   initial state + float32 actions + per-step reference outputs
  fn_occ200.npz       direct call of _get_observation with d_sen = 1.2 so that the
                      occupied list exceeds 200 and the sensed list exceeds 80
  metrics.npz         wrapper metrics (assembly_wrapper.py:48-165) on a few states
"""
from __future__ import annotations

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(REPO, "oracle"))
sys.path.insert(0, REPO)

import ref_env  # noqa: E402
import oracle_lib as ol  # noqa: E402


def record_rollout(tag, seed, n_a, steps, warm_rule_steps=0, is_boundary=True, con_self=True,
                   act_scale=1.0):
    np.random.seed(seed)
    env = ref_env.make_ref_env(n_a=n_a, is_boundary=is_boundary, is_con_self_state=con_self,
                               agent_strategy="rule" if warm_rule_steps else "input")
    e = env.env
    env.reset()
    for _ in range(warm_rule_steps):                 # drive agents into the shape (ENV:525-596)
        env.step(np.zeros((2, n_a), np.float32))
    e.agent_strategy = "input"
    out = dict(n_a=n_a, r_avoid=e.r_avoid, l_cell=e.l_cell, d_sen=e.d_sen, is_boundary=is_boundary,
               is_con_self_state=con_self, grid=np.ascontiguousarray(e.grid_center),
               p0=e.p.copy(), dp0=e.dp.copy(), nbr0=e.neighbor_index.copy(), obs0=e.obs.copy(),
               in_flags0=e.in_flags.copy(), sensed0=e.sensed_index.copy())
    keys = ["act", "obs", "rew", "prior", "p", "dp", "nbr", "in_flags", "sensed", "occupied", "collide"]
    rec = {k: [] for k in keys}
    for _ in range(steps):
        a = (act_scale * np.random.uniform(-1, 1, (2, n_a))).astype(np.float32)
        obs, rew, done, info, prior = env.step(a)
        assert not done.any() and info.shape == (3, 1)
        for k, v in zip(keys, [a, obs, rew, prior, e.p, e.dp, e.neighbor_index, e.in_flags,
                               e.sensed_index, e.occupied_index, e.is_collide_b2b]):
            rec[k].append(np.array(v, copy=True))
    for k in keys:
        out[k] = np.stack(rec[k])
    np.savez_compressed(os.path.join(HERE, f"rollout_{tag}.npz"), **out)
    print(tag, "sensed max", (out["sensed"] >= 0).sum(-1).max(), "occupied max",
          (out["occupied"] >= 0).sum(-1).max(), "in_flags", out["in_flags"].sum(-1).max(),
          "reward sum", out["rew"].sum(), "contacts", int(out["collide"].sum()) // 2)


def record_occ200():
    """A crowded synthetic state with a large sensing radius: exercises the
    occupied > 200 subsample (CPP:223-233), never reached at d_sen = 0.4."""
    from marl_amd.shapes import load_shapes
    l_cells, grids = load_shapes()
    ref = ol.reference_fns()
    rng = np.random.default_rng(7)
    grid = np.ascontiguousarray(grids[2].T)          # (2, 534)
    n_a = 48
    idx = rng.choice(grid.shape[1], n_a, replace=False)
    p = np.ascontiguousarray(grid[:, idx] + rng.uniform(-0.02, 0.02, (2, n_a)))
    dp = rng.uniform(-0.5, 0.5, (2, n_a))
    heading = np.zeros((2, n_a))
    bpos = np.array([-2.4, 2.4, 2.4, -2.4])
    d_sen, r_avoid, l_cell = 1.2, 0.3, l_cells[2]
    obs, nbr, inf, sen, occ = ref.get_observation(p, dp, heading, bpos, grid, d_sen, r_avoid, l_cell)
    rew = ref.get_reward(p, dp, heading, np.zeros((2, n_a)), bpos, grid, nbr, inf, sen, occ, d_sen, r_avoid, l_cell)
    prior = ref.prior(p, dp, grid, nbr, d_sen, r_avoid, l_cell)
    print("occ200: occupied max", (occ >= 0).sum(1).max(), "sensed max", (sen >= 0).sum(1).max(),
          "in", inf.sum(), "rew", rew.sum())
    assert (occ >= 0).sum(1).max() == 200
    np.savez_compressed(os.path.join(HERE, "fn_occ200.npz"), p=p, dp=dp, grid=grid, d_sen=d_sen,
                        r_avoid=r_avoid, l_cell=l_cell, obs=obs, nbr=nbr, in_flags=inf, sensed=sen,
                        occupied=occ, rew=rew, prior=prior)


def record_metrics():
    rows = []
    for seed, warm in [(11, 0), (12, 120), (13, 60)]:
        np.random.seed(seed)
        env = ref_env.make_ref_env(n_a=30, agent_strategy="rule" if warm else "input")
        e = env.env
        env.reset()
        for _ in range(max(warm, 1)):
            env.step(np.random.uniform(-1, 1, (2, 30)).astype(np.float32))
        m = env.get_formation_metrics()
        rows.append(dict(p=e.p.copy(), grid=np.ascontiguousarray(e.grid_center), r_avoid=e.r_avoid,
                         collide=e.is_collide_b2b.copy(), **m))
    out = {}
    for k, r in enumerate(rows):
        for key, v in r.items():
            out[f"{key}_{k}"] = v
    out["count"] = len(rows)
    np.savez_compressed(os.path.join(HERE, "metrics.npz"), **out)
    print("metrics", [(r["coverage_rate"], r["num_collisions"]) for r in rows])


if __name__ == "__main__":
    assert ref_env.available(), "needs /root/reference and `make -C oracle ref`"
    record_rollout("n30_seed226", 226, 30, 12)
    record_rollout("n30_rule150", 21, 30, 10, warm_rule_steps=150, act_scale=0.3)
    record_rollout("n30_periodic", 5, 30, 10, is_boundary=False)
    record_rollout("n30_noself", 6, 30, 6, con_self=False)
    record_rollout("n10_seed3", 3, 10, 10)
    record_rollout("n64_rule80", 4, 64, 6, warm_rule_steps=80, act_scale=0.5)
    record_rollout("n1024_seed7", 7, 1024, 1)
    record_occ200()
    record_metrics()
