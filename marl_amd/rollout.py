"""The rollout / update loop of train_assembly.py:89-161 on the device: policy forward for all
E x n_a agents in one matmul chain, BatchedAssemblyEnv.step, DeviceReplay as the push, and a DDPG
update that follows maddpg.py:99-199.  PyTorch provides the networks and optimisers; the environment
and the replay gather are the HIP library.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F

from .batched import BatchedAssemblyEnv
from .replay import DeviceReplay
from .sharding import all_reduce_sum, reduce_episode_stats


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


class _FlatGrads:
    """All gradients of a module as views into one contiguous buffer, so that a data-parallel step is ONE
    all-reduce per optimiser step (the reference's helper, misc.py:33-38, issues one per parameter tensor)."""

    def __init__(self, module: nn.Module):
        params = [p for p in module.parameters() if p.requires_grad]
        self.flat = torch.zeros(sum(p.numel() for p in params), dtype=params[0].dtype, device=params[0].device)
        off = 0
        for p in params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def zero(self) -> None:
        self.flat.zero_()                                                  # keeps the views (zero_grad() would drop them)

    def average(self, group=None) -> None:
        all_reduce_sum(self.flat, group)
        self.flat.div_(dist.get_world_size(group))


class DDPGLearner:
    """One shared DDPG agent (agents.py:9-51; maddpg.py:264-291 builds a single agent type for this
    environment): policy, critic, their targets, Adam optimisers, Gaussian exploration.

    data_parallel=True (needs an initialised torch.distributed group; RCCL on GPUs, gloo in CPU tests): every
    rank updates its own replica on batches from its own environments; rank 0's initial weights are broadcast,
    and each optimiser step is preceded by one SUM all-reduce of that network's flattened gradient divided by
    the world size — what maddpg.py:138-139, 179-180 do when update(parallel=True), in two messages of
    ~0.73 MB per update (hidden 256) instead of sixteen.  Replicas therefore stay bit-identical."""

    def __init__(self, obs_dim: int, act_dim: int = 2, hidden_dim: int = 256, lr_actor: float = 1e-4,
                 lr_critic: float = 1e-3, gamma: float = 0.95, tau: float = 0.01, device="cuda", act_dtype=None,
                 data_parallel: bool = False, group=None, capturable: bool = False, explore_seed: int = 0):
        mk = lambda i, o, c: MLPNetwork(i, o, hidden_dim, c).to(device)
        self.policy, self.target_policy = mk(obs_dim, act_dim, True), mk(obs_dim, act_dim, True)
        self.critic, self.target_critic = mk(obs_dim + act_dim, 1, False), mk(obs_dim + act_dim, 1, False)
        self.group = group
        self.data_parallel = bool(data_parallel) and dist.is_available() and dist.is_initialized() and \
            dist.get_world_size(group) > 1
        if data_parallel and not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("data_parallel=True needs torch.distributed.init_process_group() first")
        if self.data_parallel:
            for net in (self.policy, self.critic):                         # one message per network
                flat = torch.cat([p.data.reshape(-1) for p in net.parameters()])
                if dist.get_rank(group) != 0:                              # SUM of (rank 0's weights, zeros, ...) = a broadcast
                    flat.zero_()
                all_reduce_sum(flat, group)
                off = 0
                for p in net.parameters():
                    p.data.copy_(flat[off:off + p.numel()].view_as(p)); off += p.numel()
        self.target_policy.load_state_dict(self.policy.state_dict())       # agents.py:39-40
        self.target_critic.load_state_dict(self.critic.state_dict())
        self.policy_optimizer = torch.optim.Adam(self.policy.parameters(), lr_actor, capturable=capturable)
        self.critic_optimizer = torch.optim.Adam(self.critic.parameters(), lr_critic, capturable=capturable)
        self._pol_grads = _FlatGrads(self.policy) if self.data_parallel else None
        self._cri_grads = _FlatGrads(self.critic) if self.data_parallel else None
        self.gamma, self.tau = gamma, tau
        self.act_dtype = act_dtype                                         # rollout inference precision; None = float32
        self._row_bufs = {}                                                # agent-major copies of the observation frames (act)
        self.niter = 0
        # bf16 rollouts on a HIP device go through the library's matrix-core kernel (marl_amd/policy_kernel.py) when the
        # network has a shape it takes; the weights are re-packed after every update
        self.fused_policy, self._policy_version, self._packed_version = None, 0, None
        self.explore_seed, self._explore_step = int(explore_seed), 0      # the kernel's exploration draws: (seed, call counter)
        if act_dtype is torch.bfloat16 and torch.device(device).type == "cuda" and hidden_dim in (32, 64, 128, 256) and act_dim <= 32:
            from .policy_kernel import FusedPolicyBF16
            self.fused_policy = FusedPolicyBF16(self.policy)

    @torch.no_grad()
    def act(self, obs: torch.Tensor, out: torch.Tensor, noise_scale: float = 0.0, epsilon: float = 0.0,
            generator: Optional[torch.Generator] = None, env0: int = 0, explore_step: Optional[int] = None) -> torch.Tensor:
        """agents.py:69-96 for a batch of environments: obs (E, obs_dim, n_a) -> out (E, 2, n_a).
        Per environment (the reference draws its coin once per step of its one environment, agents.py:87):
        with probability epsilon the whole team acts uniformly at random, else Gaussian noise is added and the
        action clamped to [-1, 1].  generator: a torch.Generator on the environment's device; without one, the
        matrix-core policy kernel (bf16 rollouts) draws the exploration itself, keyed by explore_seed, the global
        environment id env0 + e and a call counter, and writes the final actions in one launch (explore_step: the
        counter to use instead of the learner's own, for callers that act on several shards of one time step)."""
        E, d, n = obs.shape
        if self.fused_policy is not None and obs.is_contiguous() and out.is_contiguous() and \
                (generator is None or (noise_scale <= 0.0 and epsilon <= 0.0)):
            self._refresh_fused()
            if explore_step is None:
                self._explore_step += 1
                explore_step = self._explore_step
            return self.fused_policy.forward(obs, out, noise_scale=max(noise_scale, 0.0), epsilon=max(epsilon, 0.0),
                                             seed=self.explore_seed, step=explore_step, env0=env0)
        if self.fused_policy is not None and obs.is_contiguous():
            a = self._fused(obs, torch.empty_like(out)).transpose(1, 2)    # (E, n, act_dim), as below
        else:
            x = self._agent_rows(obs)                                      # maddpg.py:84: observations.t()
            if self.act_dtype is not None:                                 # e.g. torch.bfloat16: MFMA inference for rollouts
                with torch.autocast("cuda", dtype=self.act_dtype):
                    a = self.policy(x).float()
            else:
                a = self.policy(x)
            a = a.view(E, n, -1)
        if noise_scale > 0.0:                                              # agents.py:91-94
            a = (a + noise_scale * torch.randn(a.shape, device=a.device, generator=generator)).clamp_(-1.0, 1.0)
        if epsilon > 0.0:                                                  # agents.py:87-89, one coin per environment
            coin = torch.rand(E, 1, 1, device=a.device, generator=generator) < epsilon
            a = torch.where(coin, torch.empty_like(a).uniform_(-1.0, 1.0, generator=generator), a)
        out.copy_(a.transpose(1, 2))
        return out

    def _agent_rows(self, obs: torch.Tensor) -> torch.Tensor:
        """(E, obs_dim, n_a) frames -> (E * n_a, obs_dim) rows, one per agent.  On the device through the library's tiled
        transpose (asm_obs_rows_f32) into a buffer kept per shape; elsewhere a strided copy."""
        E, d, n = obs.shape
        if not (obs.is_cuda and obs.is_contiguous() and obs.dtype is torch.float32):
            return obs.transpose(1, 2).reshape(E * n, d)
        from . import _lib
        key = (E, d, n, obs.device)
        buf = self._row_bufs.get(key)
        if buf is None:
            buf = self._row_bufs[key] = torch.empty(E * n, d, dtype=torch.float32, device=obs.device)
        _lib.check(_lib.load().asm_obs_rows_f32(E, n, d, obs.data_ptr(), buf.data_ptr(),
                                                torch.cuda.current_stream(obs.device).cuda_stream), "asm_obs_rows_f32")
        return buf

    def policy_changed(self) -> None:
        """Tell the learner that the policy's weights were changed by something Python did not see run (a replayed HIP
        graph); in-place changes made from Python (optimiser steps, load_state_dict) are noticed by themselves."""
        self._policy_version += 1

    def _refresh_fused(self) -> None:
        # the packed bf16 copy follows the float32 weights: tensors count their in-place modifications (_version)
        key = (self._policy_version, sum(p._version for p in self.policy.parameters()))
        if key != self._packed_version:
            self.fused_policy.repack(); self._packed_version = key

    def _fused(self, obs, out):
        self._refresh_fused()
        return self.fused_policy.forward(obs, out)

    # maddpg.py:99-190 in the pieces a data-parallel replica runs between its two gradient exchanges
    # (maddpg.py:138-139, 179-180): critic loss + backward | critic step, policy loss + backward | policy step.
    def critic_backward(self, obs, acs, rews, next_obs, dones):
        """maddpg.py:120-136: the TD loss of the critic and its gradients.  Returns the loss (0-d tensor)."""
        if self.data_parallel:
            self._cri_grads.zero()
        else:
            self.critic_optimizer.zero_grad()                              # maddpg.py:120
        with torch.no_grad():                                              # maddpg.py:123-127 (.detach() at :134)
            trgt_in = torch.cat((next_obs, self.target_policy(next_obs)), dim=1)
            target_value = rews + self.gamma * self.target_critic(trgt_in) * (1 - dones)
        vf_loss = F.mse_loss(self.critic(torch.cat((obs, acs), dim=1)), target_value)   # maddpg.py:130-134
        vf_loss.backward()
        return vf_loss.detach()

    def policy_backward(self, obs, acs_prior=None, alpha: float = 0.5, branch_free: bool = False):
        """maddpg.py:144-176 after the critic's step: -Q plus the prior regulariser, and the policy's gradients.
        branch_free: the mask of maddpg.py:162 as a weight instead of an index selection (no decision on the host)."""
        if self.data_parallel:
            self._pol_grads.zero()
        else:
            self.policy_optimizer.zero_grad()                              # maddpg.py:144
        pol = self.policy(obs)
        pol_loss = -self.critic(torch.cat((obs, pol), dim=1)).mean()       # maddpg.py:153-154
        if acs_prior is not None:                                          # maddpg.py:158-174
            valid = ~(acs_prior.abs() < 1e-2).all(dim=1)
            if self.data_parallel or branch_free:                          # no data-dependent branch: every rank runs the same ops
                w = valid.to(pol.dtype).unsqueeze(1)
                reg = ((pol - acs_prior) ** 2 * w).sum() / (w.sum() * pol.shape[1]).clamp_min(1.0)
                pol_loss = pol_loss + 0.3 * alpha * reg
            elif valid.any():
                pol_loss = pol_loss + 0.3 * alpha * F.mse_loss(pol[valid], acs_prior[valid])
        pol_loss.backward()
        return pol_loss.detach()

    def exchange_critic_grads(self) -> None:
        if self.data_parallel:
            self._cri_grads.average(self.group)                            # maddpg.py:138-139

    def exchange_policy_grads(self) -> None:
        if self.data_parallel:
            self._pol_grads.average(self.group)                            # maddpg.py:179-180

    def update(self, obs, acs, rews, next_obs, dones, acs_prior=None, alpha: float = 0.5, sync_losses: bool = True,
               branch_free: bool = False):
        """maddpg.py:99-190.  Returns (critic loss, policy loss): Python floats, or 0-d tensors when
        sync_losses=False (no host synchronisation inside the update loop)."""
        vf_loss = self.critic_backward(obs, acs, rews, next_obs, dones)
        self.exchange_critic_grads()
        self.critic_optimizer.step()
        pol_loss = self.policy_backward(obs, acs_prior, alpha, branch_free)
        self.exchange_policy_grads()
        self.policy_optimizer.step()
        if sync_losses:
            return float(vf_loss), float(pol_loss)
        return vf_loss, pol_loss

    @torch.no_grad()
    def update_all_targets(self):
        """maddpg.py:192-199 with the soft update of misc.py:9-19: target * (1 - tau) + source * tau."""
        for tgt, src in ((self.target_critic, self.critic), (self.target_policy, self.policy)):
            for tp, p in zip(tgt.parameters(), src.parameters()):
                tp.copy_(tp * (1.0 - self.tau) + p * self.tau)
        self.niter += 1


def run_episode(env: BatchedAssemblyEnv, learner: DDPGLearner, replay: DeviceReplay, episode_length: int,
                noise_scale: float = 0.9, epsilon: float = 0.1, generator: Optional[torch.Generator] = None):
    """train_assembly.py:99-135 for E environments: reset, episode_length steps of act -> step, the
    environment writing straight into the replay slot.  Returns the episode statistics
    (mean reward, coverage, uniformity, contacts, distance to target — summarize() of the 8-vector)."""
    if hasattr(env, "envs"):
        return _run_episode_shards(env, learner, replay, episode_length, noise_scale, epsilon, generator)
    replay.begin_episode()
    obs = env.reset()
    for _ in range(episode_length):
        a = replay.action_slot()
        learner.act(obs, a, noise_scale, epsilon, generator, env0=env.env_id_offset)
        obs = env.step(a)[0]
        replay.advance()
    return reduce_episode_stats(env.episode_stats())


def _run_episode_shards(shards, learner: DDPGLearner, replay: DeviceReplay, episode_length: int, noise_scale: float,
                        epsilon: float, generator: Optional[torch.Generator]):
    """run_episode for a StreamShards environment: every shard acts and steps on its own stream, so one shard's
    policy launch runs beside the other's environment step and the ramps and tails of the launches overlap.  The
    shards write their rows of the same replay frames; the kernel's exploration draws are keyed by global
    environment id and time step, so the episode is the one a single environment of all E would have run."""
    replay.begin_episode()
    if learner.fused_policy is not None:
        learner._refresh_fused()                                       # packed weights: now, on the current stream
    cur = torch.cuda.current_stream(shards.device)
    for s in shards.streams:
        s.wait_stream(cur)                                             # the rollout follows whatever updated the weights
    obs = shards.reset()
    for _ in range(episode_length):
        a = replay.action_slot()
        learner._explore_step += 1
        for k, (env, (first, count)) in enumerate(zip(shards.envs, shards.ranges)):
            ak = a[first:first + count]
            with torch.cuda.stream(shards.streams[k]):
                learner.act(obs[k], ak, noise_scale, epsilon, generator, env0=env.env_id_offset, explore_step=learner._explore_step)
            obs[k] = env.step(ak, shards._stream_ptrs[k])[0]
        replay.advance()
    return reduce_episode_stats(shards.episode_stats())                # (the current stream waits for the shards here)


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
    """train_assembly.py:141-157.  Returns the (critic loss, policy loss) pairs as floats; the losses are read
    back once, after the last update, so the loop itself never waits for the device."""
    if num_updates is None:
        num_updates = 30 if batch_size >= 2048 else 20
    losses = []
    for _ in range(num_updates):
        if len(replay) >= batch_size:
            obs, acs, rews, nxt, dones, prior = replay.sample(batch_size, is_prior=env.training_method == "llm_rl")
            losses.append(learner.update(obs, acs, rews, nxt, dones, prior, env.alpha, sync_losses=False))
        learner.update_all_targets()
    if losses:
        flat = torch.stack([torch.stack(pair) for pair in losses]).cpu().tolist()
        losses = [(v, p) for v, p in flat]
    return losses


class StagedUpdate:
    """One DDPG update cut into the three pieces between which data-parallel replicas exchange gradients:

        A  batch (batch_fn), critic loss + backward            -> all-reduce of the critic's flattened gradient
        B  critic step, policy loss + backward                 -> all-reduce of the policy's flattened gradient
        C  policy step, losses, soft target update, after_fn

    On a HIP device the pieces are captured into HIP graphs -- one graph for a single-rank learner, three for
    data-parallel replicas, the all-reduces issued eagerly between their replays, so nothing of the collective
    library is ever captured -- and replayed per update.  Without graphs (use_graphs=False: CPU tests, gloo
    rehearsals) the same pieces run eagerly in the same order.  batch_fn() -> (obs, acs, rews, next_obs, dones,
    prior) must write into tensors that stay where they are between calls when graphs are used."""

    def __init__(self, learner: DDPGLearner, batch_fn, alpha_fn, device, use_graphs: bool, after_fn=None):
        if use_graphs and torch.device(device).type != "cuda":
            raise RuntimeError("HIP graphs need a HIP device")
        self.learner, self.batch_fn, self.alpha_fn, self.after_fn = learner, batch_fn, alpha_fn, after_fn
        self.device, self.use_graphs = torch.device(device), bool(use_graphs)
        self.losses = torch.zeros(2, dtype=torch.float32, device=device)
        self.graphs = None
        self._batch = None
        self.launches_per_update = 0                                # graph replays + collectives of the last run (for the records)

    # the three pieces
    def _a(self):
        self._batch = self.batch_fn()
        obs, acs, rews, nxt, dones, _ = self._batch
        self.losses[0].copy_(self.learner.critic_backward(obs, acs, rews, nxt, dones))

    def _b(self):
        self.learner.critic_optimizer.step()
        obs, prior = self._batch[0], self._batch[5]
        self.losses[1].copy_(self.learner.policy_backward(obs, prior, self.alpha_fn(), branch_free=True))

    def _c(self):
        self.learner.policy_optimizer.step()
        self.learner.update_all_targets()
        if self.after_fn is not None:
            self.after_fn()

    def _groups(self):
        return [[self._a], [self._b], [self._c]] if self.learner.data_parallel else [[self._a, self._b, self._c]]

    def _eager(self):
        self._a(); self.learner.exchange_critic_grads()
        self._b(); self.learner.exchange_policy_grads()
        self._c()

    def capture(self, warmup: int):
        """Runs `warmup` updates eagerly (allocator, lazy optimiser state; they count as updates) and returns their
        losses as a (warmup, 2) tensor; then records the pieces."""
        out = torch.zeros(warmup, 2, dtype=torch.float32, device=self.device)
        if not self.use_graphs:
            for k in range(warmup):
                self._eager(); out[k].copy_(self.losses)
            self.graphs = []
            return out
        cur = torch.cuda.current_stream(self.device)
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for k in range(warmup):
                self._eager(); out[k].copy_(self.losses)
        cur.wait_stream(side)
        niter = self.learner.niter
        self.graphs, pool = [], None
        for group in self._groups():
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side, pool=pool):
                for piece in group:
                    piece()
            pool = g.pool()                                         # the pieces share tensors: one memory pool for all graphs
            self.graphs.append(g)
        self.learner.niter = niter                                  # capturing ran nothing
        return out

    def run_once(self):
        """One update; the losses are in self.losses afterwards."""
        if not self.use_graphs:
            self._eager()
            self.launches_per_update = 0
            return
        exch = (self.learner.exchange_critic_grads, self.learner.exchange_policy_grads)
        for k, g in enumerate(self.graphs):
            g.replay()
            if len(self.graphs) > 1 and k < 2:
                exch[k]()
        self.learner.niter += 1
        self.launches_per_update = len(self.graphs) + (2 if len(self.graphs) > 1 else 0)


class GraphedTrainPhase:
    """train_assembly.py:141-157 with one update (index draw, gather, DDPG update, soft target update) captured into
    HIP graphs and replayed num_updates times: the update is a few dozen small kernels, and launched one by one from
    Python the 30 updates of an episode take longer than the 200 environment steps before them.

    The captured draw reads its counter from a device word that is incremented inside the graph, so every replay
    samples new transitions; the table of complete episode slots it reads is a device tensor the replay keeps up to
    date outside the graph.  The graphs are re-captured when the set of complete slots changes (the first episodes,
    until the replay is full).  Needs optimisers built with capturable=True (DDPGLearner(..., capturable=True));
    losses come back as a (num_updates, 2) tensor without host round trips in between.

    Data-parallel replicas (DDPGLearner(data_parallel=True)): the update is cut at its two gradient exchanges into three
    graphs, and the flattened all-reduces run eagerly between their replays (StagedUpdate) -- five launches per update
    instead of one, nothing RCCL-specific captured."""

    def __init__(self, env: BatchedAssemblyEnv, learner: DDPGLearner, replay: DeviceReplay, batch_size: int = 2048):
        self.env, self.learner, self.replay, self.N = env, learner, replay, int(batch_size)
        dev = env.device
        self.counter = torch.zeros(1, dtype=torch.int32, device=dev)
        self.idx = torch.empty(self.N, dtype=torch.int64, device=dev)
        self.batch = None
        self.is_prior = env.training_method == "llm_rl"
        self.staged = StagedUpdate(learner, self._batch, lambda: self.env.alpha, dev, use_graphs=True,
                                   after_fn=lambda: self.counter.add_(1))
        self.captured_key = None

    @property
    def losses(self):
        return self.staged.losses

    @property
    def graph(self):                                                # (the first of the recorded graphs; None before the first run)
        return self.staged.graphs[0] if self.staged.graphs else None

    def _batch(self):
        self.replay.draw_indices(self.N, out=self.idx, counter_dev=self.counter)
        self.batch = self.replay.gather(self.idx, self.is_prior, out=self.batch)
        return self.batch

    def run(self, num_updates=None):
        """Returns the (num_updates, 2) tensor of (critic loss, policy loss)."""
        if num_updates is None:
            num_updates = 30 if self.N >= 2048 else 20
        if len(self.replay) < self.N:
            for _ in range(num_updates):
                self.learner.update_all_targets()                   # train_assembly.py:147-157: targets move regardless
            return torch.zeros(0, 2)
        out = torch.empty(num_updates, 2, dtype=torch.float32, device=self.env.device)
        done = 0
        key = tuple(self.replay.complete)
        if self.staged.graphs is None or self.captured_key != key:
            done = min(3, num_updates)
            draws0 = self.replay.draws
            out[:done] = self.staged.capture(done)
            self.replay.draws = draws0 + done + 1                   # the capture's own call of the draw ran nothing
            self.captured_key = key
        for k in range(done, num_updates):
            self.staged.run_once()
            out[k].copy_(self.staged.losses)
        self.learner.policy_changed()                               # the replays moved the weights behind Python's back
        return out
