"""Config 5: the train_assembly.py-shaped loop (train_assembly.py:89-161) with the environment, the replay storage
and the learner on the GPU — rollout-phase agent-steps/s and end-to-end episodes/s, one JSON line from rank 0.
(bench.py keeps the headline metric; this is the next row's measurement.)

    python tools/bench_rollout.py [--envs 4096] [--episodes 3] [--episode-length 200]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/bench_rollout.py ...

N > 1: one rank per GPU; every rank rolls out its own shard of global environment ids into its own replay and
updates its own learner replica on its own batches, gradients averaged by one flattened RCCL all-reduce per
optimiser step (DDPGLearner(data_parallel=True)); the episode statistics are all-reduced once per episode.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
if "--tunableop" in sys.argv:                                              # read by PyTorch when it loads
    os.environ["PYTORCH_TUNABLEOP_ENABLED"] = "1"
    os.environ.setdefault("PYTORCH_TUNABLEOP_FILENAME", "/tmp/tunableop_results.csv")
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from marl_amd.batched import BatchedAssemblyEnv  # noqa: E402
from marl_amd.replay import DeviceReplay  # noqa: E402
from marl_amd.rollout import DDPGLearner, run_episode, train_phase  # noqa: E402
from marl_amd.sharding import shard_range, summarize  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=4096, help="environments per GPU")
ap.add_argument("--agents", type=int, default=30)
ap.add_argument("--episodes", type=int, default=3)
ap.add_argument("--episode-length", type=int, default=200)
ap.add_argument("--slots", type=int, default=2)
ap.add_argument("--batch-size", type=int, default=2048)
ap.add_argument("--graph", action="store_true", help="train phase replayed from HIP graphs (GraphedTrainPhase): one graph per update, or three with the gradient all-reduces between them when the learner is data-parallel")
ap.add_argument("--act-dtype", choices=["float32", "bfloat16"], default="float32", help="rollout inference precision")
ap.add_argument("--shards", type=int, default=1, help="step the rank's environments as this many shards on streams of their own (StreamShards)")
ap.add_argument("--no-fused", action="store_true", help="bf16 rollouts through PyTorch autocast instead of the library's matrix-core policy kernel")
ap.add_argument("--strided-copy", action="store_true", help="float32 rollouts: agent rows by PyTorch's strided copy of the transposed frames "
                "instead of the library's tiled transpose (the previous behaviour, for before / after lines)")
ap.add_argument("--tunableop", action="store_true", help="PyTorch TunableOp: the GEMM kernels are chosen by timing at first use (float32 rollouts gain ~6 %)")
a = ap.parse_args()

world = int(os.environ.get("WORLD_SIZE", "1"))
rank = int(os.environ.get("RANK", "0"))
local_rank = int(os.environ.get("LOCAL_RANK", "0"))
# rehearsal on a one-GPU box: every rank on device 0, gloo instead of RCCL (which refuses two ranks on one device)
if os.environ.get("MARL_AMD_BENCH_ONE_DEVICE") == "1":
    local_rank = 0
backend = os.environ.get("MARL_AMD_BENCH_BACKEND", "nccl")
torch.cuda.set_device(local_rank)
device = torch.device("cuda", local_rank)
if world > 1:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)

torch.manual_seed(rank)
if a.shards > 1:
    from marl_amd.sharding import StreamShards
    env = StreamShards(a.envs, a.agents, parts=a.shards, device=device, env_id_offset=shard_range(rank, world, a.envs)[0])
else:
    env = BatchedAssemblyEnv(a.envs, a.agents, device=device, env_id_offset=shard_range(rank, world, a.envs)[0])
rp = DeviceReplay(env, a.episode_length, a.slots, seed=rank)
learner = DDPGLearner(env.obs_dim_agent, device=device, act_dtype=torch.bfloat16 if a.act_dtype == "bfloat16" else None,
                      data_parallel=world > 1, capturable=a.graph)
if a.no_fused:
    learner.fused_policy = None
if a.strided_copy:
    learner._agent_rows = lambda obs: obs.transpose(1, 2).reshape(obs.shape[0] * obs.shape[2], obs.shape[1])
graphed = None
if a.graph:
    from marl_amd.rollout import GraphedTrainPhase
    graphed = GraphedTrainPhase(env, learner, rp, a.batch_size)
run_episode(env, learner, rp, a.episode_length if a.episode_length < 5 else 5)      # warm-up: kernels loaded, allocator primed
rp.slot, rp.t, rp.filled_slots = -1, 0, 0
gather_s = train_time = roll_time = 0.0
train_each = []
stats = None
for ep in range(a.episodes):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    stats = summarize(run_episode(env, learner, rp, a.episode_length), a.agents)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    if graphed is not None:
        graphed.run()
    else:
        train_phase(env, learner, rp, a.batch_size)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    for _ in range(30):                                                # the sampler alone, same count as the phase above
        rp.sample(a.batch_size)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    roll_time += t1 - t0; train_time += t2 - t1; gather_s += t3 - t2
    train_each.append(t2 - t1)
if world > 1:
    t = torch.tensor([roll_time, train_time, gather_s], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    roll_time, train_time, gather_s = t.tolist()
steps = a.episodes * a.episode_length
if rank == 0:
    print(json.dumps({
        "workload": f"{a.envs} envs x {a.agents} agents per GPU x {world} GPU(s), {a.episodes} episodes of {a.episode_length} steps, "
                    f"policy/critic MLP 256, 30 updates of {a.batch_size} per episode and rank, replay in HBM "
                    f"({a.slots} episode slots), rollout inference in {a.act_dtype}"
                    + (" (fused matrix-core kernel)" if learner.fused_policy is not None else "")
                    + (" -- lower precision than the reference's float32 policy (networks.py:22-44), not comparable" if a.act_dtype != "float32" else "")
                    + (", agent rows by PyTorch's strided copy" if a.strided_copy and a.act_dtype == "float32" else "")
                    + (", GEMM kernels chosen by TunableOp" if a.tunableop else "")
                    + (f", {a.shards} shards per GPU on streams of their own" if a.shards > 1 else "")
                    + (", learner replicas with averaged gradients" if world > 1 else "")
                    + (", train phase replayed from a HIP graph" if a.graph else ""),
        "n_gpus": world,
        "rollout_agent_steps_per_s": world * a.envs * a.agents * steps / roll_time,
        "rollout_ms_per_step": roll_time / steps * 1e3,
        "train_phase_s_per_episode": train_time / a.episodes,
        "train_phase_s_each_episode": train_each,                     # the first episodes include HIP-graph captures (--graph)
        "sampler_s_per_episode_30_batches": gather_s / a.episodes,
        "env_episodes_per_s_end_to_end": world * a.envs * a.episodes / (roll_time + train_time),
        "replay_bytes_per_gpu": sum(t.numel() * 4 for t in (rp.obs_ring, rp.act_ring, rp.prior_ring, rp.rew_ring) if t is not None),
        "last_episode": stats}), flush=True)
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
