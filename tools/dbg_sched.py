"""Debug helper: first mismatching sensed-index row between k_step_fast schedules and the generic kernel."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from marl_amd.batched import BatchedAssemblyEnv
E, n_a = 21, 30
res = {}
for T in sys.argv[1:] or ["0", "1"]:
    from marl_amd import _lib
    _lib.load().asm_set_step_schedule(int(T))
    env = BatchedAssemblyEnv(E, n_a, seed=11, keep_indices=True)
    env.reset()
    rng = np.random.default_rng(4)
    grid = env.grid_center.cpu().numpy(); n_g = env.n_g.cpu().numpy()
    p = env.p.cpu().numpy()
    for e in range(0, E, 2):
        p[e] = grid[e][:, rng.choice(n_g[e], n_a, replace=False)] + rng.normal(0, 0.02, (2, n_a))
    env.p.copy_(torch.from_numpy(p)); env.observe()
    rec = []
    for t in range(4):
        a = torch.from_numpy(rng.uniform(-1, 1, (E, 2, n_a)).astype(np.float32)).cuda()
        env.step(a)
        rec.append((env.sensed_index.cpu().numpy().copy(), env.obs.cpu().numpy().copy(), env.reward.cpu().numpy().copy()))
    res[T] = rec
ref = res[list(res)[0]]
for T, rec in res.items():
    for t, (s, o, r) in enumerate(rec):
        bad = np.argwhere((s != ref[t][0]).any(-1))
        print(f"T={T} step {t}: sensed rows differing {len(bad)}, obs equal {np.array_equal(o, ref[t][1])}, reward equal {np.array_equal(r, ref[t][2])}")
        for e, i in bad[:2]:
            a, b = ref[t][0][e, i], s[e, i]
            k = int(np.argmax(a != b))
            print(f"   env {e} agent {i}: count ref {(a >= 0).sum()} got {(b >= 0).sum()}; first diff at slot {k}: ref {a[max(0,k-2):k+6]} got {b[max(0,k-2):k+6]}")
