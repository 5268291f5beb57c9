"""Replay storage for the assembly environments.

Two classes:

* ``ReplayBufferAgent`` — the reference's host-side buffer
  (marl_llm/algorithm/utils/buffer_agent.py:13-198) restated over torch tensors on any device,
  same constructor, ``push``/``sample``/``get_average_rewards`` signatures and the same quirks
  (tail overwrite instead of wrap-around, sampling window that ignores the fill level).  It lets
  train_assembly.py:83-155 run unchanged against the single-environment mirror.

* ``DeviceReplay`` — the batched counterpart: episode slots in HBM into which the step kernel
  writes observations, rewards and priors directly (no push copy; an observation is stored once
  and serves as next_obs of the previous step), and a HIP gather kernel that returns
  agent-major batches like buffer_agent.py:157-177.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import _lib


class ReplayBufferAgent:
    """buffer_agent.py:13-198 with torch storage (float64 like the numpy original by default)."""

    BEGIN_INDEX_RANGE = 300000                                  # buffer_agent.py:145

    def __init__(self, max_steps, num_agents, start_stop_index, state_dim, action_dim, device="cpu",
                 dtype=torch.float64):
        self.max_steps = max_steps
        self.num_agents = num_agents
        self.total_length = self.max_steps * self.num_agents       # buffer_agent.py:45
        z = lambda d: torch.zeros(self.total_length, d, dtype=dtype, device=device)
        self.obs_buffs = z(state_dim)
        self.ac_buffs = z(action_dim)
        self.ac_prior_buffs = z(action_dim)
        self.log_pi_buffs = z(1)
        self.rew_buffs = z(1)
        self.next_obs_buffs = z(state_dim)
        self.done_buffs = z(1)
        self.filled_i = 0
        self.curr_i = 0
        self.agent_index = start_stop_index
        self.device = torch.device(device)
        self.dtype = dtype

    def __len__(self):
        return self.filled_i

    def _rows(self, x, index):
        """x[:, index].T as a tensor of the storage dtype on the storage device."""
        t = torch.as_tensor(np.asarray(x) if not torch.is_tensor(x) else x)
        return t[:, index].T.to(device=self.device, dtype=self.dtype)

    def push(self, observations_orig, actions_orig, rewards_orig, next_observations_orig, dones_orig, index,
             actions_prior_orig=None, log_pi_orig=None):
        """buffer_agent.py:67-128."""
        data_length = len(range(index.start, index.stop))
        if self.curr_i + data_length > self.total_length:          # buffer_agent.py:102-104: overwrite the tail
            rollover = data_length - (self.total_length - self.curr_i)
            self.curr_i -= rollover
        sl = slice(self.curr_i, self.curr_i + data_length)
        self.obs_buffs[sl] = self._rows(observations_orig, index)
        self.ac_buffs[sl] = self._rows(actions_orig, index)
        self.rew_buffs[sl] = self._rows(rewards_orig, index)
        self.next_obs_buffs[sl] = self._rows(next_observations_orig, index)
        self.done_buffs[sl] = self._rows(dones_orig, index)
        if actions_prior_orig is not None:
            self.ac_prior_buffs[sl] = self._rows(actions_prior_orig, index)
        if log_pi_orig is not None:
            self.log_pi_buffs[sl] = self._rows(log_pi_orig, index)
        self.curr_i += data_length
        if self.filled_i < self.total_length:                      # buffer_agent.py:123-124
            self.filled_i += data_length
        if self.curr_i == self.total_length:
            self.curr_i = 0

    def sample(self, N, to_gpu=False, is_prior=False, is_log_pi=False):
        """buffer_agent.py:130-177.  Index draws use numpy's global generator like the original, so a
        seeded run picks the same rows.  Returns float32 tensors (torch.Tensor(...) in the original)."""
        begin_index = np.random.randint(0, self.BEGIN_INDEX_RANGE)
        inds = np.random.choice(
            np.arange(begin_index, self.total_length - self.BEGIN_INDEX_RANGE + begin_index, dtype=np.int32),
            size=N, replace=False)
        idx = torch.from_numpy(inds.astype(np.int64)).to(self.device)
        dev = torch.device("cuda") if to_gpu else torch.device("cpu")
        cast = lambda b: b[idx].to(device=dev, dtype=torch.float32).requires_grad_(False)
        return (cast(self.obs_buffs), cast(self.ac_buffs), cast(self.rew_buffs), cast(self.next_obs_buffs),
                cast(self.done_buffs), cast(self.ac_prior_buffs) if is_prior else None,
                cast(self.log_pi_buffs) if is_log_pi else None)

    def get_average_rewards(self, N):
        """buffer_agent.py:179-198 (rows, not agents, are indexed by the original; kept as is)."""
        if self.filled_i == self.max_steps:
            inds = np.arange(self.curr_i - N, self.curr_i)
        else:
            inds = np.arange(max(0, self.curr_i - N), self.curr_i)
        r = self.rew_buffs.cpu().numpy()
        return [r[i][inds].mean() for i in range(self.num_agents)]


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


class DeviceReplay:
    """Episode slots in HBM for a BatchedAssemblyEnv.

        replay = DeviceReplay(env, episode_length=200, slots=4)
        replay.begin_episode(); obs = env.reset()
        for t in range(200):
            a = replay.action_slot()            # (E, 2, n_a) tensor to fill with the policy's actions
            ...; env.step(a); replay.advance()
        batch = replay.sample(2048)             # obs, act, rew, next_obs, done, a_prior

    begin_episode() and advance() point the environment's obs / reward / a_prior outputs at the next
    frame of the current slot, so stepping is the push (train_assembly.py:123-125 without the copy).
    sample() draws from complete slots only: the slot being written (which, when reused, mixes frames of two
    episodes) is left out until its last step.
    """

    def __init__(self, env, episode_length: int, slots: int, seed: int = 0):
        self.env = env
        self.lib = _lib.load()
        self.T, self.S = int(episode_length), int(slots)
        E, n, d, dev = env.E, env.n_a, env.obs_dim_agent, env.device
        f32 = dict(dtype=torch.float32, device=dev)
        self.obs_ring = torch.zeros(self.S, self.T + 1, E, d, n, **f32)
        self.act_ring = torch.zeros(self.S, self.T, E, 2, n, **f32)
        self.prior_ring = torch.zeros(self.S, self.T, E, 2, n, **f32) if env.training_method == "llm_rl" else None
        self.rew_ring = torch.zeros(self.S, self.T, E, 1, n, **f32)
        self.slot, self.t = -1, 0
        self.complete = [False] * self.S                            # per slot: holds one whole episode
        self.seed, self.draws = int(seed), 0
        # the complete slots, in order, as a device table the index draw renumbers through: kept up to date here, outside
        # any HIP graph that captured a draw, so a replayed draw always lands in slots that hold whole episodes
        self._slot_table = torch.zeros(self.S, dtype=torch.int64, device=dev)

    # ---- writing ---------------------------------------------------------------------
    def begin_episode(self) -> None:
        """Call before env.reset(): the reset's observation becomes frame 0 of the next slot.  From here until
        its last step the slot is not sampled (it mixes frames of two episodes); an episode given up early
        (begin_episode() again before episode_length steps) simply leaves its slot out until it is rewritten."""
        self.slot = (self.slot + 1) % self.S
        self.t = 0
        self.complete[self.slot] = False
        self._refresh_slot_table()
        self.env.obs = self.obs_ring[self.slot, 0]

    def action_slot(self) -> torch.Tensor:
        """Where this step's actions are stored; pass it to env.step() after filling it."""
        self._point_outputs()
        return self.act_ring[self.slot, self.t]

    def _point_outputs(self) -> None:
        e = self.env
        e.obs = self.obs_ring[self.slot, self.t + 1]
        e.reward = self.rew_ring[self.slot, self.t]
        if self.prior_ring is not None:
            e.a_prior = self.prior_ring[self.slot, self.t]

    def advance(self) -> None:
        """Call after env.step()."""
        self.t += 1
        if self.t == self.T:
            self.complete[self.slot] = True
            self._refresh_slot_table()

    def _refresh_slot_table(self) -> None:
        slots = [k for k in range(self.S) if self.complete[k]]
        slots += [slots[-1] if slots else 0] * (self.S - len(slots))
        self._slot_table.copy_(torch.tensor(slots, dtype=torch.int64))

    @property
    def filled_slots(self) -> int:
        """Complete episodes available to sample()."""
        return sum(self.complete)

    @filled_slots.setter
    def filled_slots(self, n: int) -> None:                         # tests fill the rings by hand
        self.complete = [k < n for k in range(self.S)]
        self._refresh_slot_table()

    def __len__(self) -> int:
        """Stored transitions (agent rows), as len(ReplayBufferAgent)."""
        return self.filled_slots * self.T * self.env.E * self.env.n_a

    # ---- sampling --------------------------------------------------------------------
    def draw_indices(self, N: int, out: Optional[torch.Tensor] = None, counter_dev: Optional[torch.Tensor] = None,
                     without_replacement: bool = False) -> torch.Tensor:
        """N transition ids, uniform over the complete slots.  counter_dev: a device int32 word added to the draw
        counter inside the kernel (for launches captured into a HIP graph: increment it between replays); `out`:
        where to write the ids.  By default the ids are drawn with replacement (one launch, nothing read back).
        without_replacement=True gives N distinct ids like buffer_agent.py:149-154 (np.random.choice(...,
        replace=False)): duplicates within the batch are drawn again until none is left, which reads one flag back
        per round and is therefore not for captured draws."""
        if len(self) == 0:
            raise RuntimeError("no completed episode to sample from")
        if without_replacement and counter_dev is not None:
            raise ValueError("a draw without replacement decides on the host: it cannot be captured into a HIP graph")
        if without_replacement and N > len(self):
            raise ValueError(f"cannot draw {N} distinct ids out of {len(self)}")      # np.random.choice raises as well
        idx = torch.empty(N, dtype=torch.int64, device=self.env.device) if out is None else out

        def draw(dst):
            with torch.cuda.device(self.env.device):
                rc = self.lib.asm_replay_draw_indices(self.seed, self.draws, _ptr(counter_dev), 0, len(self), dst.numel(),
                                                      _ptr(dst), self.env._stream())
            _lib.check(rc, "asm_replay_draw_indices")
            self.draws += 1

        draw(idx)
        if without_replacement:
            while True:                                             # expected duplicates: N^2 / (2 len): ~0.1 per batch of 2048 out of 2.4e7
                srt, order = torch.sort(idx)
                dup = torch.zeros(N, dtype=torch.bool, device=idx.device)
                dup[order[1:]] = srt[1:] == srt[:-1]                # every repeat after the first of its value
                n_dup = int(dup.sum())
                if n_dup == 0:
                    break
                fresh = torch.empty(n_dup, dtype=torch.int64, device=idx.device)
                draw(fresh)
                idx[dup] = fresh
        slots = [k for k in range(self.S) if self.complete[k]]
        if counter_dev is not None or slots != list(range(len(slots))):
            # a slot in between is being rewritten: renumber through the table of complete slots (always, for a captured
            # draw: the table is read when the graph is replayed)
            per_slot = self.T * self.env.E * self.env.n_a
            remapped = self._slot_table[idx // per_slot] * per_slot + idx % per_slot
            idx = idx.copy_(remapped)
        return idx

    def gather(self, idx: torch.Tensor, is_prior: bool = True, out=None):
        """Rows of the transitions idx (int64 ids on the device): the tuple of buffer_agent.py:175-177
        without log_pi: (obs, act, rew, next_obs, done, a_prior).  out: a tuple from an earlier call to write into."""
        e = self.env
        if idx.dtype != torch.int64 or not idx.is_contiguous() or idx.device != self.obs_ring.device:
            raise ValueError("idx must be a contiguous int64 tensor on the environment's device")
        N, d = idx.numel(), e.obs_dim_agent
        f32 = dict(dtype=torch.float32, device=e.device)
        want_prior = is_prior and self.prior_ring is not None
        if out is not None:
            obs, act, rew, nxt, done, pri = out
        else:
            obs, nxt = torch.empty(N, d, **f32), torch.empty(N, d, **f32)
            act, rew = torch.empty(N, 2, **f32), torch.empty(N, 1, **f32)
            pri = torch.empty(N, 2, **f32) if want_prior else None
            done = torch.zeros(N, 1, **f32)                         # assembly.py:475-477: never done
        with torch.cuda.device(e.device):
            rc = self.lib.asm_replay_gather_f32(e.E, e.n_a, d, self.T, _ptr(self.obs_ring), _ptr(self.act_ring),
                                                _ptr(self.prior_ring), _ptr(self.rew_ring), _ptr(idx), N, _ptr(obs),
                                                _ptr(act), _ptr(rew), _ptr(nxt), _ptr(pri), e._stream())
        _lib.check(rc, "asm_replay_gather_f32")
        return obs, act, rew, nxt, done, pri

    def sample(self, N: int, is_prior: bool = True, without_replacement: bool = False):
        return self.gather(self.draw_indices(N, without_replacement=without_replacement), is_prior)
