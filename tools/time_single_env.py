"""Step time of the single-environment, reference-shaped path (five C symbols, host buffers): marl_amd.gym mirror."""
import os, sys, time
from types import SimpleNamespace
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
from marl_amd import gym
from marl_amd.gym.wrappers import AssemblySwarmWrapper
from marl_amd.cfg import AssemblyArgs
args = AssemblyArgs()
env = AssemblySwarmWrapper(gym.make("AssemblySwarm-v0").unwrapped, args)
np.random.seed(226)
env.reset()
n = env.num_agents
for _ in range(20):
    env.step(np.random.uniform(-1, 1, (2, n)).astype(np.float32))
t = time.perf_counter(); K = 200
for _ in range(K):
    env.step(np.random.uniform(-1, 1, (2, n)).astype(np.float32))
dt = (time.perf_counter() - t) / K
print(f"single env, n_a={n}: {dt * 1e3:.3f} ms per step = {n / dt:.0f} agent-steps/s")
