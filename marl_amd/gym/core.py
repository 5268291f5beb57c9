"""Minimal Env / Wrapper / registry with the behaviour the assembly scripts rely on
(reference: cus_gym/gym/core.py:8-273, cus_gym/gym/envs/registration.py:60-110,185):
`make(id)` adds a TimeLimit when the spec has max_episode_steps, `.unwrapped` strips
wrappers, and Wrapper forwards unknown public attributes to the wrapped env."""
from __future__ import annotations

import importlib


class Env:
    metadata = {"render.modes": []}
    reward_range = (-float("inf"), float("inf"))
    spec = None
    action_space = None
    observation_space = None

    def step(self, action):
        raise NotImplementedError

    def reset(self):
        raise NotImplementedError

    def render(self, mode="human"):
        raise NotImplementedError

    def close(self):
        pass

    def seed(self, seed=None):
        return

    @property
    def unwrapped(self):
        return self

    def __str__(self):
        return f"<{type(self).__name__} instance>" if self.spec is None else f"<{type(self).__name__}<{self.spec.id}>>"


class Wrapper(Env):
    def __init__(self, env):
        self.env = env
        self.action_space = env.action_space
        self.observation_space = env.observation_space
        self.reward_range = env.reward_range
        self.metadata = env.metadata

    def __getattr__(self, name):                      # core.py:232-237
        if name.startswith("_"):
            raise AttributeError(f"attempted to get missing private attribute '{name}'")
        return getattr(self.env, name)

    @property
    def spec(self):
        return self.env.spec

    def step(self, action):
        return self.env.step(action)

    def reset(self, **kwargs):
        return self.env.reset(**kwargs)

    def render(self, mode="human", **kwargs):
        return self.env.render(mode, **kwargs)

    def close(self):
        return self.env.close()

    @property
    def unwrapped(self):
        return self.env.unwrapped


class TimeLimit(Wrapper):
    """Episode cut-off after max_episode_steps (wrappers/time_limit.py of gym 0.19)."""

    def __init__(self, env, max_episode_steps=None):
        super().__init__(env)
        self._max_episode_steps = max_episode_steps
        self._elapsed_steps = None

    def step(self, action):
        assert self._elapsed_steps is not None, "Cannot call env.step() before calling reset()"
        out = list(self.env.step(action))
        self._elapsed_steps += 1
        if self._elapsed_steps >= self._max_episode_steps:
            if isinstance(out[3], dict):
                out[3]["TimeLimit.truncated"] = True
            out[2] = True
        return tuple(out)

    def reset(self, **kwargs):
        self._elapsed_steps = 0
        return self.env.reset(**kwargs)


class EnvSpec:
    def __init__(self, id, entry_point, max_episode_steps=None, reward_threshold=None, kwargs=None):
        self.id = id
        self.entry_point = entry_point
        self.max_episode_steps = max_episode_steps
        self.reward_threshold = reward_threshold
        self._kwargs = dict(kwargs or {})

    def make(self, **kwargs):
        kw = dict(self._kwargs); kw.update(kwargs)
        ep = self.entry_point
        if not callable(ep):
            mod, attr = ep.split(":")
            ep = getattr(importlib.import_module(mod), attr)
        env = ep(**kw)
        env.unwrapped.spec = self
        return env


registry = {}


def register(id, **kwargs):
    if id in registry:
        raise ValueError(f"Cannot re-register id: {id}")
    registry[id] = EnvSpec(id, **kwargs)


def make(id, **kwargs):
    if id not in registry:
        raise KeyError(f"No registered env with id: {id}")
    spec = registry[id]
    env = spec.make(**kwargs)
    if spec.max_episode_steps is not None:
        env = TimeLimit(env, max_episode_steps=spec.max_episode_steps)
    return env
