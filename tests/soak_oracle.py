"""Soak against the CPU oracle: many environments, teacher-forced step by step (the oracle is fed the device's own
float32 state, as in tests/gpu_util.py), in the reference's reset regime and with the agents packed inside their
shapes.  Indices bit-exact, values to one float32 rounding, like the GPU tests, but over many more agent-steps.

    python tests/soak_oracle.py [E] [steps] [n_a]      (not collected by pytest; part of the checker side, like the tests)
"""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "oracle"), os.path.dirname(os.path.abspath(__file__))]
import torch  # noqa: E402
from marl_amd.batched import BatchedAssemblyEnv  # noqa: E402
from gpu_util import cfg_from_env, teacher_forced_step  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
n_a = int(sys.argv[3]) if len(sys.argv) > 3 else 30

for regime in ("random", "packed"):
    env = BatchedAssemblyEnv(E, n_a, seed=12, keep_indices=True)
    cfg = cfg_from_env(env)
    env.reset()
    if regime == "packed":
        g = torch.Generator(device="cuda").manual_seed(4)
        cell = (torch.rand(E, n_a, device="cuda", generator=g) * env.n_g[:, None]).long()
        pos = torch.gather(env.grid_center, 2, cell[:, None, :].expand(E, 2, n_a))
        env.p.copy_(pos + 0.03 * torch.randn(E, 2, n_a, device="cuda", generator=g))
        env.observe()
    rng = np.random.default_rng(7)
    t0 = time.time(); inside = 0
    for t in range(steps):
        act = rng.uniform(-1, 1, (E, 2, n_a)).astype(np.float32)
        teacher_forced_step(env, cfg, act, f"{regime} step {t}")
        inside += int(env.in_flags.sum())
        if t % 10 == 9:
            print(f"{regime}: {t + 1} steps x {E} envs match the oracle ({time.time() - t0:.0f} s)", flush=True)
    print(f"{regime}: {steps * E * n_a} agent-steps checked, inside fraction {inside / (steps * E * n_a):.3f}, "
          f"sub-sampled lists at the last step {int((env.sensed_index[..., -1] >= 0).sum())}", flush=True)
