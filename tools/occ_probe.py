import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from marl_amd.sharding import StreamShards
cap = int(sys.argv[1])
sh = StreamShards(4096, 30, parts=2, seed=226, num_obs_grid_max=cap)
sh.reset()
acts = sh.on_each(lambda k, env: torch.stack([env.random_actions(step=j) for j in range(32)]))
for t in range(1500):
    sh.step([a[t % 32] for a in acts])
sh.synchronize()
t0 = time.perf_counter()
n = 3000
for t in range(n):
    sh.step([a[t % 32] for a in acts])
sh.synchronize()
print(f"cap {cap}: {(time.perf_counter() - t0) / n * 1e6:.2f} us per step, lib {os.environ.get('MARL_AMD_LIB', 'default')}")
