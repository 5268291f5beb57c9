"""The rollout / update loop of train_assembly.py:89-161 on the device: policy forward for all
E x n_a agents in one matmul chain, BatchedAssemblyEnv.step, DeviceReplay as the push, and a DDPG
update that follows maddpg.py:99-199.  PyTorch provides the networks and optimisers; the environment
and the replay gather are the HIP library.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .batched import BatchedAssemblyEnv
from .replay import DeviceReplay
from .sharding import reduce_episode_stats


class MLPNetwork(nn.Module):
    """networks.py:6-44: four linear layers, leaky_relu, tanh on the output of a policy."""

    def __init__(self, input_dim: int, out_dim: int, hidden_dim: int = 64, constrain_out: bool = False):
        super().__init__()
        self.fc1 = nn.Linear(input_dim, hidden_dim)
        self.fc2 = nn.Linear(hidden_dim, hidden_dim)
        self.fc3 = nn.Linear(hidden_dim, hidden_dim)
        self.fc4 = nn.Linear(hidden_dim, out_dim)
        self.constrain_out = constrain_out

    def forward(self, x):
        h = F.leaky_relu(self.fc1(x))
        h = F.leaky_relu(self.fc2(h))
        h = F.leaky_relu(self.fc3(h))
        out = self.fc4(h)
        return torch.tanh(out) if self.constrain_out else out


class DDPGLearner:
    """One shared DDPG agent (agents.py:9-51; maddpg.py:264-291 builds a single agent type for this
    environment): policy, critic, their targets, Adam optimisers, Gaussian exploration."""

    def __init__(self, obs_dim: int, act_dim: int = 2, hidden_dim: int = 256, lr_actor: float = 1e-4,
                 lr_critic: float = 1e-3, gamma: float = 0.95, tau: float = 0.01, device="cuda", act_dtype=None):
        mk = lambda i, o, c: MLPNetwork(i, o, hidden_dim, c).to(device)
        self.policy, self.target_policy = mk(obs_dim, act_dim, True), mk(obs_dim, act_dim, True)
        self.critic, self.target_critic = mk(obs_dim + act_dim, 1, False), mk(obs_dim + act_dim, 1, False)
        self.target_policy.load_state_dict(self.policy.state_dict())       # agents.py:39-40
        self.target_critic.load_state_dict(self.critic.state_dict())
        self.policy_optimizer = torch.optim.Adam(self.policy.parameters(), lr_actor)
        self.critic_optimizer = torch.optim.Adam(self.critic.parameters(), lr_critic)
        self.gamma, self.tau = gamma, tau
        self.act_dtype = act_dtype                                         # rollout inference precision; None = float32
        self.niter = 0

    @torch.no_grad()
    def act(self, obs: torch.Tensor, out: torch.Tensor, noise_scale: float = 0.0, epsilon: float = 0.0,
            generator: Optional[torch.Generator] = None) -> torch.Tensor:
        """agents.py:69-96 for a batch of environments: obs (E, obs_dim, n_a) -> out (E, 2, n_a).
        With probability epsilon the whole step is uniform random, else Gaussian noise is added and
        the action clamped to [-1, 1]."""
        E, d, n = obs.shape
        x = obs.transpose(1, 2).reshape(E * n, d)                          # maddpg.py:84: observations.t()
        if self.act_dtype is not None:                                     # e.g. torch.bfloat16: MFMA inference for rollouts
            with torch.autocast("cuda", dtype=self.act_dtype):
                a = self.policy(x).float()
        else:
            a = self.policy(x)
        if epsilon > 0.0 and torch.rand((), generator=generator, device="cpu").item() < epsilon:
            a = torch.empty_like(a).uniform_(-1.0, 1.0, generator=generator)
        elif noise_scale > 0.0:
            a = (a + noise_scale * torch.randn(a.shape, device=a.device, generator=generator)).clamp_(-1.0, 1.0)
        out.copy_(a.view(E, n, -1).transpose(1, 2))
        return out

    def update(self, obs, acs, rews, next_obs, dones, acs_prior=None, alpha: float = 0.5):
        """maddpg.py:99-190."""
        self.critic_optimizer.zero_grad()
        with torch.no_grad():
            trgt_in = torch.cat((next_obs, self.target_policy(next_obs)), dim=1)
            target_value = rews + self.gamma * self.target_critic(trgt_in) * (1 - dones)
        vf_loss = F.mse_loss(self.critic(torch.cat((obs, acs), dim=1)), target_value)
        vf_loss.backward()
        self.critic_optimizer.step()

        self.policy_optimizer.zero_grad()
        pol = self.policy(obs)
        pol_loss = -self.critic(torch.cat((obs, pol), dim=1)).mean()
        if acs_prior is not None:                                          # maddpg.py:158-174
            valid = ~(acs_prior.abs() < 1e-2).all(dim=1)
            if valid.any():
                pol_loss = pol_loss + 0.3 * alpha * F.mse_loss(pol[valid], acs_prior[valid])
        pol_loss.backward()
        self.policy_optimizer.step()
        return float(vf_loss.detach()), float(pol_loss.detach())

    @torch.no_grad()
    def update_all_targets(self):
        """maddpg.py:192-199 (soft update, misc.py)."""
        for tgt, src in ((self.target_critic, self.critic), (self.target_policy, self.policy)):
            for tp, p in zip(tgt.parameters(), src.parameters()):
                tp.mul_(1.0 - self.tau).add_(p, alpha=self.tau)
        self.niter += 1


def run_episode(env: BatchedAssemblyEnv, learner: DDPGLearner, replay: DeviceReplay, episode_length: int,
                noise_scale: float = 0.9, epsilon: float = 0.1, generator: Optional[torch.Generator] = None):
    """train_assembly.py:99-135 for E environments: reset, episode_length steps of act -> step, the
    environment writing straight into the replay slot.  Returns the episode statistics
    (mean reward, coverage, uniformity, contacts, distance to target — summarize() of the 8-vector)."""
    replay.begin_episode()
    obs = env.reset()
    for _ in range(episode_length):
        a = replay.action_slot()
        learner.act(obs, a, noise_scale, epsilon, generator)
        obs = env.step(a)[0]
        replay.advance()
    return reduce_episode_stats(env.episode_stats())


@torch.no_grad()
def eval_episode(env: BatchedAssemblyEnv, learner: DDPGLearner, episode_length: int = 300,
                 shape_schedule: Optional[dict] = None):
    """eval_assembly.py:115-190 for E environments: one reset, then `episode_length` noise-free steps; before
    step t the environments switch to shape `shape_schedule[t]` when t is a key (eval_assembly.py:149-153,
    process_shape), and the formation metrics of the current state are recorded (:155-170).
    Returns (metrics (episode_length, E, 5) float64 on the device, columns BatchedAssemblyEnv.METRIC_FIELDS,
    total reward per environment (E,))."""
    obs = env.reset()
    metrics = torch.empty(episode_length, env.E, 5, dtype=torch.float64, device=env.device)
    total = torch.zeros(env.E, dtype=torch.float64, device=env.device)
    a = torch.empty(env.E, 2, env.n_a, dtype=torch.float32, device=env.device)
    for t in range(episode_length):
        if shape_schedule and t in shape_schedule:
            env.set_shape(shape_schedule[t])
        env.formation_metrics(metrics[t])
        learner.act(obs, a)                                                # explore=False: no noise (:123, :174)
        obs, rew = env.step(a)[:2]
        total += rew.sum(dim=(1, 2), dtype=torch.float64)
    return metrics, total


def collect_expert_episode(env: BatchedAssemblyEnv, replay: DeviceReplay, episode_length: int):
    """eval/collect_expert_data.py:58-77 for E environments: one episode driven by the in-environment 'rule'
    expert (assembly.py:525-596), every transition (obs, expert action, next obs) landing in the replay slot the
    environment steps into.  The environment must keep its index lists (keep_indices=True).  Returns the episode
    statistics vector, reduced over the ranks."""
    replay.begin_episode()
    env.reset()
    for _ in range(episode_length):
        a = replay.action_slot()
        env.rule_actions(out=a)
        env.step(a)
        replay.advance()
    return reduce_episode_stats(env.episode_stats())


def train_phase(env: BatchedAssemblyEnv, learner: DDPGLearner, replay: DeviceReplay, batch_size: int = 2048,
                num_updates: Optional[int] = None):
    """train_assembly.py:141-157."""
    if num_updates is None:
        num_updates = 30 if batch_size >= 2048 else 20
    losses = []
    for _ in range(num_updates):
        if len(replay) >= batch_size:
            obs, acs, rews, nxt, dones, prior = replay.sample(batch_size, is_prior=env.training_method == "llm_rl")
            losses.append(learner.update(obs, acs, rews, nxt, dones, prior, env.alpha))
        learner.update_all_targets()
    return losses
