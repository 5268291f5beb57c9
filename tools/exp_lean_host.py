"""Is the Python side of env.step() what limits three or four stream shards?  Steps S shards of E / S environments for
n steps (a) through StreamShards.step and (b) through argument lists built beforehand (one ctypes call per launch,
nothing else on the host), same kernels, same streams; no episode ends inside.

    python tools/exp_lean_host.py [E] [n_a] [S] [n]
"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402
from marl_amd.sharding import StreamShards  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n_a = int(sys.argv[2]) if len(sys.argv) > 2 else 30
S = int(sys.argv[3]) if len(sys.argv) > 3 else 3
n = int(sys.argv[4]) if len(sys.argv) > 4 else 190

sh = StreamShards(E, n_a, parts=S, seed=226, keep_indices=False)
ring = 64
acts = sh.on_each(lambda k, env: torch.stack([env.random_actions(step=j) for j in range(ring)]))


def timed(fn):
    sh.reset(0)
    for t in range(5):
        sh.step([a[t % ring] for a in acts])
    sh.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(S)]
    for k in range(S):
        ev[k][0].record(sh.streams[k])
    t0 = time.perf_counter()
    fn()
    host = time.perf_counter() - t0
    for k in range(S):
        ev[k][1].record(sh.streams[k])
    sh.synchronize()
    return max(ev[k][0].elapsed_time(ev[k][1]) for k in range(S)) * 1e3 / n, host * 1e6 / n


def through_python():
    for t in range(n):
        sh.step([a[t % ring] for a in acts])


# argument lists of every (step, shard) beforehand
sh.reset(0)
sh.step([a[0] for a in acts]); sh.synchronize()            # builds each environment's cached list
lists = []
for t in range(ring):
    row = []
    for k, env in enumerate(sh.envs):
        args = list(env._step_args)
        args[4] = acts[k][t].data_ptr(); args[17] = 0
        args[18] = env._lat_ref if env.lattice_valid else None
        args[19] = sh.streams[k].cuda_stream
        row.append(tuple(args))
    lists.append(row)
f = sh.envs[0].lib.asm_batch_step_f32


def lean():
    for t in range(n):
        for args in lists[t % ring]:
            f(*args)


for rep in range(3):
    a = timed(through_python)
    b = timed(lean)
    print(f"S={S}: through StreamShards.step {a[0]:.2f} us per step (host {a[1]:.2f}); bare ctypes calls {b[0]:.2f} us per step (host {b[1]:.2f})")
