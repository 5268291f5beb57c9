"""Multi-GPU glue of the batched environment: environments are independent, so ranks
own contiguous ranges of GLOBAL environment ids (the reset generator is keyed by that
id, so any split yields the same environments) and the only collective is one SUM
all-reduce of the 8-value episode-statistics vector per episode (RCCL over xGMI on
GPUs; the same code runs on gloo for CPU tests)."""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist

STATS_FIELDS = ("reward_sum", "reward_sq_sum", "coverage_sum", "uniformity_sum", "contact_pairs_sum",
                "avg_distance_sum", "env_steps", "agents_inside_sum")


def shard_range(rank: int, world: int, envs_per_rank: int) -> Tuple[int, int]:
    """(first global env id, count) of `rank` for weak scaling: every rank owns envs_per_rank ids."""
    if not (0 <= rank < world) or envs_per_rank < 1:
        raise ValueError("bad shard request")
    return rank * envs_per_rank, envs_per_rank


def split_range(total_envs: int, rank: int, world: int) -> Tuple[int, int]:
    """(first id, count) for strong scaling: `total_envs` ids split as evenly as possible."""
    base, rem = divmod(total_envs, world)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def all_reduce_sum(t: torch.Tensor, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """SUM all-reduce of t in place.  RCCL reduces the device tensor where it lies; under gloo (CPU tests, and
    rehearsals of the multi-rank control flow with several ranks on one GPU) a device tensor goes through the
    host, because gloo's own device-tensor path stalls when ranks share a busy GPU."""
    if t.is_cuda and dist.get_backend(group) == "gloo":
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def reduce_episode_stats(local_stats: torch.Tensor, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """Sum the per-rank [8] float64 vector over all ranks (in place) and return it."""
    if local_stats.shape != (8,) or local_stats.dtype != torch.float64:
        raise ValueError("episode statistics are 8 float64 values")
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        all_reduce_sum(local_stats, group)
    return local_stats


def summarize(stats: torch.Tensor, n_a: int) -> dict:
    """Episode means as TRAIN:164-197 logs them, from the reduced vector."""
    s = stats.tolist()
    steps = max(s[6], 1.0)
    mean_r = s[0] / (steps * n_a)
    return {"reward_mean": mean_r, "reward_std": max(s[1] / (steps * n_a) - mean_r ** 2, 0.0) ** 0.5,
            "coverage": s[2] / steps, "uniformity": s[3] / steps, "collisions_per_step": s[4] / steps,
            "avg_distance_to_target": s[5] / steps, "inside_fraction": s[7] / (steps * n_a)}


class StreamShards:
    """One GPU's environments as `parts` independent BatchedAssemblyEnv shards, each on its own HIP stream.

    A single launch over all environments leaves the chip part-empty while its workgroups ramp up and
    while the last ones drain; with two launches in flight, one shard's ramp and tail run under the other
    shard's steady state (4096 environments on one MI355X: 58 -> 46 us per step of the whole batch).
    The shards own contiguous global environment ids, so together they are exactly the environments of
    one BatchedAssemblyEnv(num_envs, env_id_offset=...).  Each shard is driven on its stream; nothing
    orders the shards against each other until synchronize() or episode_stats().
    """

    def __init__(self, num_envs: int, n_a: int = 30, *, parts: int = 2, device="cuda", env_id_offset: int = 0, **env_kw):
        from .batched import BatchedAssemblyEnv
        if parts < 1 or num_envs < parts:
            raise ValueError("need at least one environment per shard")
        self.device = torch.device(device)
        self.parts, self.E, self.n_a = parts, int(num_envs), int(n_a)
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(parts)]
        self._stream_ptrs = [s.cuda_stream for s in self.streams]
        self.envs = []
        first = 0
        for k in range(parts):
            _, count = split_range(self.E, k, parts)
            with torch.cuda.stream(self.streams[k]):
                self.envs.append(BatchedAssemblyEnv(count, n_a, device=self.device, env_id_offset=env_id_offset + first, **env_kw))
            first += count

    # ---- the shards seen as one environment of E (DeviceReplay, the train phase, run_episode) ----
    @property
    def ranges(self):
        """(first environment, count) of every shard within this object's E environments."""
        out, first = [], 0
        for env in self.envs:
            out.append((first, env.E)); first += env.E
        return out

    obs_dim_agent = property(lambda self: self.envs[0].obs_dim_agent)
    training_method = property(lambda self: self.envs[0].training_method)
    alpha = property(lambda self: self.envs[0].alpha)
    env_id_offset = property(lambda self: self.envs[0].env_id_offset)

    def _stream(self):
        import ctypes as C
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _split(self, name: str, t: torch.Tensor) -> None:
        for env, (a, n) in zip(self.envs, self.ranges):
            setattr(env, name, t[a:a + n])                            # rows of environments: a contiguous block each

    # Assigning an (E, ...) tensor points every shard's output at its rows of it (what DeviceReplay does per step).
    obs = property(lambda self: [env.obs for env in self.envs], lambda self, t: self._split("obs", t))
    reward = property(lambda self: [env.reward for env in self.envs], lambda self, t: self._split("reward", t))
    a_prior = property(lambda self: [env.a_prior for env in self.envs], lambda self, t: self._split("a_prior", t))

    def on_each(self, fn, only: Optional[int] = None):
        """fn(k, env) for every shard (or shard `only`) with the shard's stream current; returns the list of results."""
        out = []
        for k, env in enumerate(self.envs):
            if only is None or only == k:
                with torch.cuda.stream(self.streams[k]):
                    out.append(fn(k, env))
        return out

    def reset(self, episode=None):
        return self.on_each(lambda k, env: env.reset(episode))

    def step(self, actions):
        """actions: one (E_k, 2, n_a) tensor per shard, produced on that shard's stream.  The launches name
        their stream explicitly, so no stream context is entered on the host."""
        return [env.step(actions[k], self._stream_ptrs[k]) for k, env in enumerate(self.envs)]

    def synchronize(self) -> None:
        for s in self.streams:
            s.synchronize()

    def capture(self, fn) -> "ShardGraph":
        """Record fn() -- any sequence of step()/on_each() calls whose tensors stay alive -- into one HIP graph
        that spans the shard streams.  Replaying it costs one launch on the host instead of one per shard and
        step (with two shards of 2048 environments the host keeps up anyway and a replay runs at the same speed).  The environment step allocates nothing and never synchronises, so it can
        be captured as it is."""
        cur = torch.cuda.current_stream(self.device)
        cap = torch.cuda.Stream(device=self.device)
        cap.wait_stream(cur)
        for s in self.streams:
            cap.wait_stream(s)
        steps0 = [env.step_count for env in self.envs]
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=cap, capture_error_mode="thread_local"):
            for s in self.streams:                                # fork: the shard streams join the capture
                s.wait_stream(cap)
            fn()
            for s in self.streams:                                # join
                cap.wait_stream(s)
        cur.wait_stream(cap)
        steps = [env.step_count - s0 for env, s0 in zip(self.envs, steps0)]
        for env, s0 in zip(self.envs, steps0):                    # capturing runs nothing
            env.step_count = s0
        return ShardGraph(self, graph, steps)

    def episode_stats(self) -> torch.Tensor:
        """Sum of the shards' 8-value vectors, on the current stream (which waits for the shards)."""
        cur = torch.cuda.current_stream(self.device)
        parts = self.on_each(lambda k, env: env.episode_stats())
        for s in self.streams:
            cur.wait_stream(s)
        return torch.stack(parts).sum(dim=0)


class ShardGraph:
    """A captured sequence of shard steps (StreamShards.capture).  replay() launches it on the current stream,
    ordered after everything already queued on the shard streams; work queued on the shard streams afterwards
    runs after the replay."""

    def __init__(self, shards: StreamShards, graph, steps):
        self.shards, self.graph, self.steps = shards, graph, steps

    def replay(self) -> None:
        sh = self.shards
        cur = torch.cuda.current_stream(sh.device)
        for s in sh.streams:
            cur.wait_stream(s)
        self.graph.replay()
        for s in sh.streams:
            s.wait_stream(cur)
        for env, n in zip(sh.envs, self.steps):
            env.step_count += n
