"""Soak: the multi-wave step kernel against the generic kernel, bit for bit, over many environments and
steps in both regimes (reference reset; agents packed inside their shapes).  Rare paths (exact
re-evaluations, sub-sampling across queue refills, the reward's double-precision fallbacks) get hit here.

    python tools/soak.py [E] [steps]
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402
from marl_amd.batched import BatchedAssemblyEnv  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300


def make(T):
    from marl_amd import _lib
    _lib.load().asm_set_step_schedule(int(T))
    return BatchedAssemblyEnv(E, 30, seed=5, keep_indices=True)


def pack(env, g):
    cell = (torch.rand(E, 30, device="cuda", generator=g) * env.n_g[:, None]).long()
    pos = torch.gather(env.grid_center, 2, cell[:, None, :].expand(E, 2, 30))
    env.p.copy_(pos + 0.03 * torch.randn(E, 2, 30, device="cuda", generator=g))
    env.observe()


bad = 0
for regime in ("random", "packed"):
    a, b = make("0"), make("4")
    a.reset(); b.reset()
    if regime == "packed":
        pack(a, torch.Generator(device="cuda").manual_seed(1)); pack(b, torch.Generator(device="cuda").manual_seed(1))
    rew = inside = 0.0
    for t in range(steps):
        act = a.random_actions(step=t) * (0.3 if regime == "packed" else 1.0)
        ra, rb = a.step(act), b.step(act)
        for name, x, y in (("obs", ra[0], rb[0]), ("reward", ra[1], rb[1]), ("prior", ra[4], rb[4]), ("p", a.p, b.p), ("dp", a.dp, b.dp),
                           ("nbr", a.neighbor_index, b.neighbor_index), ("in_flags", a.in_flags, b.in_flags),
                           ("sensed", a.sensed_index, b.sensed_index), ("occupied", a.occupied_index, b.occupied_index)):
            if not torch.equal(x, y):
                bad += 1
                print(f"{regime} step {t}: {name} differs in {int((x != y).sum())} elements")
        rew += float(ra[1].sum()); inside += float(a.in_flags.sum())
        if bad > 5:
            sys.exit(1)
    same = torch.allclose(a.stats, b.stats, rtol=1e-9, atol=1e-12)
    print(f"{regime}: {steps} steps x {E} envs identical={bad == 0}, stats close={same}, mean reward {rew / (steps * E * 30):.4f}, "
          f"inside fraction {inside / (steps * E * 30):.3f}, sub-sampled agent-steps {int((a.sensed_index[:, :, -1] >= 0).sum())} at the last step")
sys.exit(1 if bad else 0)
