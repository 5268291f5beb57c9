"""Distribution of the sensed-list sizes of a batch after some random steps (plain torch arithmetic on the state the
environment holds; single precision, so a handful of borderline cells may differ from the kernel's exact decisions --
this is a workload description, not a check).

    python tools/list_sizes.py [E] [n_a] [steps]

Prints, per agent: the share inside the shape, the share whose list is empty / shorter than the cap / sub-sampled, the
mean list length; per wave of 64 consecutive agents: how many lanes sub-sample, the longest list, the largest number of
list bits one slot consumes.
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402
from marl_amd.batched import BatchedAssemblyEnv  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n_a = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 60
env = BatchedAssemblyEnv(E, n_a)
env.reset()
cap = int(env.params.num_obs_grid_max)
d_sen = float(env.params.d_sen)
for t in range(steps):
    env.step(env.random_actions())
torch.cuda.synchronize()
n_after_all, inflag_all = [], []
for e in range(min(E, 64)):
    n_g = int(env.n_g[e])
    g = env.grid_center[e, :, :n_g].double()                       # [2, n_g]
    p = env.p[e].double()                                          # [2, n_a]
    l = float(env.l_cell[e])
    d2 = (g[0][None, :] - p[0][:, None]) ** 2 + (g[1][None, :] - p[1][:, None]) ** 2   # [n_a, n_g]
    sensed = d2 < d_sen * d_sen
    occ = (d2 <= (l / 2) ** 2).any(0)                              # a cell some agent sits in
    inflag = d2.min(1).values.sqrt() < np.sqrt(2.0) * l / 2
    n_before = sensed.sum(1)
    n_occ = (sensed & occ[None, :]).sum(1)
    n_after = torch.where(inflag, n_before - n_occ, n_before)
    n_after_all.append(n_after.cpu().numpy()); inflag_all.append(inflag.cpu().numpy())
n = np.stack(n_after_all); f = np.stack(inflag_all)
print(f"E sampled {n.shape[0]}, agents {n_a}, cap {cap}, d_sen {d_sen}, after {steps} random steps")
print(f"inside the shape {f.mean():.3f}; list empty {np.mean(n == 0):.3f}, 1..cap {np.mean((n > 0) & (n <= cap)):.3f}, "
      f"sub-sampled {np.mean(n > cap):.3f}; mean length {n.mean():.1f}, of the non-empty {n[n > 0].mean():.1f}, max {n.max()}")
print(f"sub-sampled among agents outside the shape {np.mean(n[~f] > cap):.3f}, inside {np.mean(n[f] > cap):.3f}; "
      f"mean length outside {n[~f].mean():.1f}, inside {n[f].mean():.1f}")
w = n.reshape(n.shape[0], -1, 64)
sub = (w > cap).sum(2)
print("lanes per wave that sub-sample: mean %.1f, histogram (0,1,2,3-4,5-8,9-16,17+): %s" % (
    sub.mean(), np.histogram(sub, bins=[0, 1, 2, 3, 5, 9, 17, 65])[0].tolist()))
print("longest list per wave: mean %.1f; p10/p50/p90 %s" % (w.max(2).mean(), np.percentile(w.max(2), [10, 50, 90]).tolist()))
print("wave-mean slots in use (sum of min(n,cap) / 64): %.1f of %d; wave-max %.1f" % (
    np.minimum(w, cap).mean(), cap, np.minimum(w, cap).max(2).mean()))
dmax = np.ceil((np.maximum(w.max(2), cap) - 1) / (cap - 1))
print("largest bits per slot in a wave: histogram (1,2,3,4+): %s" % np.histogram(dmax, bins=[1, 2, 3, 4, 99])[0].tolist())
