"""Would a sixth workgroup per CU pay?  Steps 4096 x 30 in three stream shards with a shorter cell-list capacity
(num_obs_grid_max = 48: 25.7 KB of LDS per workgroup instead of 29.8 KB, so LDS admits six workgroups per CU) and prints
the step time; run it once with the library as built (96 registers: five workgroups per CU) and once with a build for
six waves per SIMD (make with -DASM_FAST_MINW_LAT=6: 80 registers, spills) -- same workload, same LDS, only the
residency differs.

    MARL_AMD_LIB=<lib> python tools/exp_six_resident.py [cap] [E] [S]
"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402
from marl_amd.sharding import StreamShards  # noqa: E402

cap = int(sys.argv[1]) if len(sys.argv) > 1 else 48
E = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
S = int(sys.argv[3]) if len(sys.argv) > 3 else 3
n_a, n, ring = 30, 190, 64
sh = StreamShards(E, n_a, parts=S, seed=226, keep_indices=False, num_obs_grid_max=cap)
acts = sh.on_each(lambda k, env: torch.stack([env.random_actions(step=j) for j in range(ring)]))
sh.synchronize()
best = None
for rep in range(6):
    sh.reset(0)
    for t in range(5):
        sh.step([a[t % ring] for a in acts])
    sh.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(S)]
    for k in range(S):
        ev[k][0].record(sh.streams[k])
    for t in range(n):
        sh.step([a[t % ring] for a in acts])
    for k in range(S):
        ev[k][1].record(sh.streams[k])
    sh.synchronize()
    us = max(ev[k][0].elapsed_time(ev[k][1]) for k in range(S)) * 1e3 / n
    best = us if best is None or us < best else best
print(f"cap {cap}, {E} x {n_a}, {S} shards, lib {os.path.basename(os.environ.get('MARL_AMD_LIB', 'default'))}: {best:.2f} us per step "
      f"(best of 6 x {n} steps), schedule {sh.envs[0].step_schedule()}")
