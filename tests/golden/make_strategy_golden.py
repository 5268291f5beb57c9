"""Records tests/golden/strategy_<name>.npz: the reference Python environment stepping itself with its
in-environment policies (agent_strategy 'rule' and 'llm', assembly.py:520-596) — initial state and the
state / observation / reward after every step.  Run in the build container:

    make -C oracle ref && python tests/golden/make_strategy_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(REPO, "oracle"))
sys.path.insert(0, REPO)
import ref_env  # noqa: E402

for name, seed, steps in (("rule", 31, 40), ("llm", 32, 25)):
    np.random.seed(seed)
    env = ref_env.make_ref_env(n_a=30, agent_strategy=name)
    e = env.env
    env.reset()
    out = dict(n_a=30, r_avoid=e.r_avoid, l_cell=e.l_cell, grid=np.ascontiguousarray(e.grid_center), p0=e.p.copy(),
               dp0=e.dp.copy(), is_boundary=True, is_con_self_state=True)
    rec = {k: [] for k in ("p", "dp", "obs", "rew", "in_flags")}
    for _ in range(steps):
        obs, rew, _, _, _ = env.step(np.zeros((2, 30), np.float32))          # the action argument is ignored
        for k, v in (("p", e.p), ("dp", e.dp), ("obs", obs), ("rew", rew), ("in_flags", e.in_flags)):
            rec[k].append(np.array(v, copy=True))
    out.update({k: np.stack(v) for k, v in rec.items()})
    np.savez_compressed(os.path.join(HERE, f"strategy_{name}.npz"), **out)
    print(name, "inside at the end:", int(out["in_flags"][-1].sum()), "reward sum", float(out["rew"].sum()))
