"""Config 5: the train_assembly.py-shaped loop with the environment, the replay storage and the
learner on one MI355X — rollout-phase agent-steps/s and end-to-end episodes/s, printed as one
JSON line.  (bench.py keeps the headline metric; this is the next row's measurement.)

    python tools/bench_rollout.py [--envs 4096] [--episodes 3] [--episode-length 200]
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402
from marl_amd.batched import BatchedAssemblyEnv  # noqa: E402
from marl_amd.replay import DeviceReplay  # noqa: E402
from marl_amd.rollout import DDPGLearner, run_episode, train_phase  # noqa: E402
from marl_amd.sharding import summarize  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=4096)
ap.add_argument("--agents", type=int, default=30)
ap.add_argument("--episodes", type=int, default=3)
ap.add_argument("--episode-length", type=int, default=200)
ap.add_argument("--slots", type=int, default=2)
ap.add_argument("--batch-size", type=int, default=2048)
ap.add_argument("--act-dtype", choices=["float32", "bfloat16"], default="float32", help="rollout inference precision")
a = ap.parse_args()

torch.manual_seed(0)
env = BatchedAssemblyEnv(a.envs, a.agents)
rp = DeviceReplay(env, a.episode_length, a.slots)
learner = DDPGLearner(env.obs_dim_agent, act_dtype=torch.bfloat16 if a.act_dtype == "bfloat16" else None)
run_episode(env, learner, rp, min(5, a.episode_length))          # warm-up: kernels loaded, allocator primed
rp.slot, rp.filled_slots = -1, 0
train_time = roll_time = 0.0
stats = None
for ep in range(a.episodes):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    stats = summarize(run_episode(env, learner, rp, a.episode_length), a.agents)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    train_phase(env, learner, rp, a.batch_size)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    roll_time += t1 - t0; train_time += t2 - t1
steps = a.episodes * a.episode_length
print(json.dumps({
    "workload": f"{a.envs} envs x {a.agents} agents, {a.episodes} episodes of {a.episode_length} steps, policy/critic MLP 256, "
                f"30 updates of {a.batch_size} per episode, replay in HBM ({a.slots} episode slots), rollout inference in {a.act_dtype}",
    "rollout_agent_steps_per_s": a.envs * a.agents * steps / roll_time,
    "rollout_ms_per_step": roll_time / steps * 1e3,
    "train_phase_s_per_episode": train_time / a.episodes,
    "env_episodes_per_s_end_to_end": a.envs * a.episodes / (roll_time + train_time),
    "replay_bytes": sum(t.numel() * 4 for t in (rp.obs_ring, rp.act_ring, rp.prior_ring, rp.rew_ring) if t is not None),
    "last_episode": stats}))
