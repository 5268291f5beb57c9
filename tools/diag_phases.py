"""Where a k_step workgroup spends its cycles: runs the DIAGNOSTIC build of the library
(make -C marl_amd/csrc diag; in-kernel s_memtime stamps) and prints the mean number of
shader cycles between consecutive stamps.  Read the SHARES, not the totals: the stamps
cost cycles of their own.

    MARL_AMD_LIB=marl_amd/csrc/build/libAssemblyEnv_diag.so python tools/diag_phases.py [E] [n_a]
"""
import ctypes as C
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault("MARL_AMD_LIB", os.path.join(REPO, "marl_amd", "csrc", "build", "libAssemblyEnv_diag.so"))
import torch  # noqa: E402
from marl_amd.batched import BatchedAssemblyEnv  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n_a = int(sys.argv[2]) if len(sys.argv) > 2 else 30
# third argument N: N - 1 more batches of the same size are stepped on streams of their own all the time (the steady
# state of the N-shard benchmark); only the first batch's launches stamp
others = []
for j in range(1, int(sys.argv[3]) if len(sys.argv) > 3 else 1):
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        o = BatchedAssemblyEnv(E, n_a, env_id_offset=j * E)
        o.reset()
        others.append((o, o.random_actions(), side))
env = BatchedAssemblyEnv(E, n_a)
lib = env.lib
env.reset()
for t in range(30):
    env.step(env.random_actions())
grid = 8192
buf = torch.zeros(grid, 32, dtype=torch.int64, device="cuda")
assert lib.asm_diag_set_buffer(C.c_void_p(buf.data_ptr())) == 0
rc = lib.asm_diag_set_filter(C.c_void_p(env.p.data_ptr())); assert rc == 0, rc    # the other batch's launches do not stamp
acc = None
for t in range(10):
    buf.zero_()
    act = env.random_actions()
    torch.cuda.synchronize()
    if others:
        for _ in range(3):
            for o, o_act, side in others:
                o.step(o_act, side.cuda_stream)
            env.step(act)
    else:
        env.step(act)
    torch.cuda.synchronize()
    b = buf.cpu().numpy()
    b = b[b[:, 9] > 0]
    d = np.diff(b[:, :10], axis=1).astype(np.float64)
    acc = d if acc is None else acc + d
    wall = (b[:, 11] - b[:, 10]).astype(np.float64) / 100.0      # s_memrealtime ticks at 100 MHz
    span = (b[:, 11].max() - b[:, 10].min()) / 100.0
lib.asm_diag_set_buffer(None)
acc /= 10
q = np.percentile(wall, [50, 90, 99, 100]); st = (b[:, 10] - b[:, 10].min()) / 100.0
print(f"workgroup lifetime us p50/p90/p99/max = {q[0]:.1f}/{q[1]:.1f}/{q[2]:.1f}/{q[3]:.1f}; start time us p50/p90/max = "
      f"{np.percentile(st, 50):.1f}/{np.percentile(st, 90):.1f}/{st.max():.1f}")
h, _ = np.histogram(st, bins=np.arange(0, max(st.max(), 1.0) + 4.0, 4.0))
print("workgroup starts per 4 us:", h.tolist())
e_ = (b[:, 11] - b[:, 10].min()) / 100.0
h, _ = np.histogram(e_, bins=np.arange(0, max(e_.max(), 1.0) + 4.0, 4.0))
print("workgroup ends per 4 us:  ", h.tolist())
if b[:, 12].any():
    if b[:, 13].all():
        x = b[:, [5, 12, 13, 14, 6, 15, 7]].astype(np.float64)
        dd = np.diff(x, axis=1)
        print("fast kernel, cycles: barrier after cell pass %.0f | P4 %.0f | barrier %.0f | P5 scan %.0f | barrier %.0f | emission %.0f" % tuple(dd.mean(0)))
    else:                                                   # lattice path: the nearest cell is settled inside P3 by the fourth wave
        x = b[:, [5, 12, 14, 6, 15, 7]].astype(np.float64)
        dd = np.diff(x, axis=1)
        print("fast kernel (lattice path), first wave's cycles: waiting at the barrier that ends P3 %.0f | up to the list phase %.0f | P5 slot lists %.0f | "
              "barrier %.0f | emission %.0f" % tuple(dd.mean(0)))
    if b[:, 16].any():
        print("P5 split: popcount pass %.0f | list loop %.0f cycles, %.1f iterations of thread 0 (max %d)" % (
            (b[:, 16] - b[:, 14]).mean(), (b[:, 6] - b[:, 16]).mean(), b[:, 17].mean(), b[:, 17].max()))
        ss = b[:, 18] > 0
        ll = (b[:, 6] - b[:, 16]).astype(np.float64)
        print("workgroups with a sub-sampling lane: %d of %d; list loop cycles %.0f with, %.0f without (%.1f iterations)" % (
            int(ss.sum()), b.shape[0], ll[ss].mean() if ss.any() else 0, ll[~ss].mean(), b[~ss, 17].mean()))
if b[:, 25].any():
    t = (b[:, [0, 19, 27, 29, 30, 28, 25, 1]] - b[:, [26]]).astype(np.float64).mean(0)
    print("P0 of the first wave, cycles after its first instruction: first stamp %.0f | n_g there %.0f | grid loads issued %.0f | shape ids there %.0f | "
          "table loads issued %.0f | first grid data %.0f | at the barrier %.0f | barrier released %.0f" % tuple(t))
if not b[:, 12].any() and b[:, 16:32].all():
    # the large swarm's kernel: when each of the 16 waves reached the statistics barrier, relative to the first wave
    rel = (b[:, 16:32].astype(np.float64) - b[:, [8]].astype(np.float64))
    print("waves at the statistics barrier, cycles after the first wave got there: mean per wave",
          [int(x) for x in rel.mean(0)], "| last wave: mean %.0f, p90 %.0f, max %.0f" % (
              rel.max(1).mean(), np.percentile(rel.max(1), 90), rel.max()))
slow = np.argsort(-wall)[:5]
print("slowest workgroups (phase cycles):", [(int(k), [int(x) for x in d[k]]) for k in slow])
order = np.argsort(wall)
mid = order[len(order) // 2 - 2:len(order) // 2 + 3]
print("median workgroups (phase cycles): ", [(int(k), [int(x) for x in d[k]]) for k in mid])
fast = order[:5]
print("fastest workgroups (phase cycles):", [(int(k), [int(x) for x in d[k]]) for k in fast])
names = ["load env -> LDS", "dynamics (forces, prior, integrate)", "contact count + state write", "k-NN + agent block",
         "cell pass", "in-flag, popcounts, target block", "sensed emission", "reward", "stats"]
if b[:, 12].any():          # the multi-wave kernel stamps the same slots at its own phase boundaries (first wave's view)
    names = ["P0 grids and state -> LDS", "P1 dynamics (first wave) / prior (second wave)", "P2 state write, two barriers",
             "P3 k-NN + agent block (first wave)", "P3 the first wave's share of the cell words",
             "barrier, P4 nearest cell, barrier, P5 slot lists", "barrier, P6 emission + reward weights",
             "P7 reward", "P7 reward statistics"]
tot = acc.sum(1).mean()
print(f"workgroups {acc.shape[0]}, mean cycles per workgroup {tot:.0f}; last launch: mean workgroup lifetime {wall.mean():.1f} us, "
      f"first start -> last end {span:.1f} us => about {wall.sum() / span / 256:.1f} workgroups resident per CU, "
      f"clock ~{tot / wall.mean() / 1e3:.2f} GHz")
for k, n in enumerate(names):
    print(f"  {n:40s} {acc[:, k].mean():10.0f}  {100 * acc[:, k].mean() / tot:5.1f} %")
