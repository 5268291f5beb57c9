"""Phase cycles of the part workgroups of a split large-swarm step (diagnostic build, see tools/diag_phases.py):
    python tools/diag_split.py [E] [n_a]"""
import ctypes as C
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault("MARL_AMD_LIB", os.path.join(REPO, "marl_amd", "csrc", "build", "libAssemblyEnv_diag.so"))
import torch  # noqa: E402
from marl_amd.batched import BatchedAssemblyEnv  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n_a = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
env = BatchedAssemblyEnv(E, n_a)
env.lib.asm_set_step_schedule(4 + 32)
env.reset()
assert env.step_schedule() & 128
for t in range(30):
    env.step(env.random_actions())
parts = (1 << (n_a - 1).bit_length()) // 256
buf = torch.zeros(E * parts, 32, dtype=torch.int64, device="cuda")
assert env.lib.asm_diag_set_buffer(C.c_void_p(buf.data_ptr())) == 0
tot = []
for t in range(10):
    buf.zero_()
    env.step(env.random_actions()); torch.cuda.synchronize()
    b = buf.cpu().numpy()[E:]                                   # rows the dynamics launch did not stamp
    d = np.diff(b[:, 3:10], axis=1).astype(np.float64)
    tot.append(d)
env.lib.asm_diag_set_buffer(None)
d = np.mean(tot, axis=0)
names = ["load + bins + k-NN + agent block", "cell pass + other parts' occupancy", "in-flag, popcounts, target block", "sensed emission", "reward", "stats"]
life = d.sum(1)
print(f"part workgroups {d.shape[0]}: cycles p10/p50/p90/max = {np.percentile(life, 10):.0f}/{np.percentile(life, 50):.0f}/{np.percentile(life, 90):.0f}/{life.max():.0f}")
for k, n in enumerate(names):
    print(f"  {n:40s} mean {d[:, k].mean():9.0f}  p90 {np.percentile(d[:, k], 90):9.0f}  max {d[:, k].max():9.0f}")
