"""Put `<repo>/marl_amd/compat` on sys.path to make `import gym` resolve to marl_amd's
assembly environment, so that scripts written against the reference's customised gym
(`import gym; from gym.wrappers import AssemblySwarmWrapper`) run unchanged."""
import sys

import marl_amd.gym as _g
from marl_amd.gym import *  # noqa: F401,F403
from marl_amd.gym import spaces, wrappers, core, envs  # noqa: F401

for _name in ("spaces", "wrappers", "core", "envs"):
    sys.modules[__name__ + "." + _name] = getattr(_g, _name)
sys.modules[__name__ + ".envs.assembly"] = __import__("marl_amd.gym.envs.assembly", fromlist=["x"])
sys.modules[__name__ + ".wrappers.assembly_wrapper"] = __import__("marl_amd.gym.wrappers.assembly_wrapper", fromlist=["x"])
