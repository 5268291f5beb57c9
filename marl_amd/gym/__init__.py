"""`gym`-shaped entry point for the assembly environment: make / Env / Wrapper /
spaces / wrappers.AssemblySwarmWrapper, as the reference scripts use them
(marl_llm/train/train_assembly.py:10-11, 55-58)."""
from . import spaces
from .core import Env, Wrapper, TimeLimit, register, make, registry

# cus_gym/gym/envs/__init__.py:14-19
register(id="AssemblySwarm-v0", entry_point="marl_amd.gym.envs.assembly:AssemblySwarmEnv",
         max_episode_steps=100, reward_threshold=1000.0)

from . import wrappers  # noqa: E402

__all__ = ["Env", "Wrapper", "TimeLimit", "register", "make", "registry", "spaces", "wrappers"]
