#!/usr/bin/env python3
"""Benchmark of the batched assembly-environment step on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--envs-per-gpu E] [--agents n_a]

One "step" = one pass of the hot path (assembly.py:482-661 for every environment of
the batch: contact/wall forces, prior, integration, observation, reward) over
E environments x n_a agents per GPU, actions already resident in HBM.
N > 1: launched by torch.distributed.run, one rank per GPU; environments are sharded
by global id, no data-path collective; one RCCL all-reduce of the 8-value
episode-statistics vector per 200-step episode.

Timing: R windows of exactly --steps steps each, back to back, bracketed by barrier +
synchronize; HIP events on the shard streams at the boundaries of blocks of windows (one
window when it spans an episode, else as many as span one) and the step time is the
median block / its steps (max over ranks).

Rank 0 prints ONE JSON line: whole-job agent-steps/s, the roofline of the dominant
kernel (k_step) derived from that value, and at N = 1 the CPU baseline (the compiled
reference AssemblyEnv.cpp if oracle/_ref travels with the tree, else our C port)
timed on the host cores in the same run.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak
HBM_COPY_GBS = 6290.0          # measured rate of a float4 copy on MI355X (79 % of the peak)
VALU_PEAK_TFLOPS = 157.3       # MI355X fp32 vector peak (256 CUs x 4 SIMDs x 16 lanes x 2 (fma) x 2 (packed) x 2.4 GHz)
EPISODE_LENGTH = 200           # marl_llm/cfg/assembly_cfg.py:192
SETTLE_STEPS = int(os.environ.get("MARL_AMD_BENCH_SETTLE", "1200"))   # untimed steps of continuous load before the step time is steady


def b_alg_per_agent_step(n_a: int, n_g_mean: float) -> float:
    """Algorithmic HBM bytes per agent-step with fp32 state and a fused step:
    reads p,dp 16 + action 8 + previous neighbour list 24 + grid share 8*n_g/n_a;
    writes p,dp 16 + obs 4*192 + reward 4 + a_prior 8 + neighbour list 24 + in_flag 4."""
    return 872.0 + 8.0 * n_g_mean / n_a


def f_alg_per_agent_step(n_a: int, n_g_mean: float) -> float:
    """Algorithmic FLOPs per agent-step (fma = 2, sqrt = 1, cos ~ 20): two nearest-cell passes 2*5*n_g, the
    environment's cell-occupancy share 5*n_g, contacts 15*(n_a-1), k-NN distances 5*(n_a-1), reward 30*80."""
    return 15.0 * n_g_mean + 20.0 * (n_a - 1) + 2400.0


def _profile_for(suffix: str, E: int, n_a: int, S: int):
    """The committed profile summary profiles/rNN_*<suffix> of the newest round taken on this workload, or None (older
    rounds' files describe older kernels)."""
    import glob
    paths = sorted(glob.glob(os.path.join(REPO, "profiles", "r[0-9][0-9]_*" + suffix)), reverse=True)
    newest = os.path.basename(paths[0])[:3] if paths else ""
    for path in (q for q in paths if os.path.basename(q).startswith(newest)):
        try:
            with open(path) as f:
                t = json.load(f)
            w = t["workload"]
            if (w["envs_per_gpu"], w["agents"]) == (E, n_a):     # (however the run shards them: the files say how they did)
                t["source"] = "profiles/" + os.path.basename(path)
                return t
        except (OSError, ValueError, KeyError, TypeError):
            pass
    return None


def measured_traffic(E: int, n_a: int, S: int):
    """HBM bytes per step-kernel launch and the issue counters from the committed rocprofv3 PMC passes
    (profiles/rNN_*pmc.json, written by tools/pmc_summary.py), if one was taken on this workload; None otherwise
    (counters cannot be read from inside this process)."""
    return _profile_for("pmc.json", E, n_a, S)


def profiled_kernel(E: int, n_a: int, S: int):
    """Name and average duration of the step kernel from the committed rocprofv3 --kernel-trace --stats run of this
    workload (profiles/rNN_*kernel_trace.json, written by tools/kernel_stats_summary.py), or None."""
    return _profile_for("kernel_trace.json", E, n_a, S)


def trace_note(kt, S: int, ms_per_step: float):
    """How the traced run compares with this one: per-dispatch tracing slows the launch pipeline, and a slower pipeline
    overlaps fewer launches, each of which then runs on a less crowded chip."""
    traced = kt.get("bench_ms_per_step_under_tracing") if kt else None
    if not traced:
        return None
    inflight = S * kt["avg_us"] * 1e-3 / traced
    if traced < 1.1 * ms_per_step:
        return f"the traced run stepped in {traced * 1e3:.1f} us ({inflight:.1f} launches in flight), like this one"
    return (f"the traced run stepped in {traced * 1e3:.1f} us with {inflight:.1f} launches in flight on average, this run in "
            f"{ms_per_step * 1e3:.1f} us: kernel_avg_us_profiled is the duration on a less crowded chip, a lower bound of the "
            f"duration here, and launch_gap_us an upper bound of the gap")


def step_kernel_name(code: int, periodic: bool, idx: int, nt: bool) -> str:
    """The kernel asm_batch_step_f32 launches, from asm_debug_step_schedule's code (include/marl_amd_env.h)."""
    T = code & 15
    if T:
        b = lambda v: "true" if v else "false"
        return f"asmenv::k_step_fast<{T}, {b(periodic)}, {idx}, {b(code & 16)}, {b(nt)}>"
    name = "asmenv::k_step<float, true>" if code & 256 else "asmenv::k_step<float>"
    extras = [x for x, on in (("lattice path", code & 32), ("cell lists", code & 64), ("three launches", code & 128)) if on]
    return name + (f" ({', '.join(extras)})" if extras else "")


def limiter_note(tr, valu_bound: bool, S: int = 1, step_us: float = 0.0) -> str:
    """What the committed counters say limits the kernel.  Counter passes serialise the launches, so the counters are
    one launch alone on the chip; the run itself keeps S launches in flight, and a step's vector-issue cycles (active VALU
    quad-cycles x 4 per SIMD, summed over the launches of a step) against the step's cycles at 2.39 GHz give the issue load
    of the run."""
    if not tr:
        return "no counter pass committed for this workload"
    vb, wa = tr.get("valu_busy"), tr.get("sq_wait_any_frac_of_wave_cycles")
    if vb is None:
        return "counter summary without VALU busy"
    head = f"one launch alone: vector issue {vb:.2f} busy" + (f", waves waiting {wa:.2f} of their cycles" if wa is not None else "")
    av = (tr.get("counters") or {}).get("SQ_ACTIVE_INST_VALU")
    per_step = (tr.get("workload") or {}).get("streams", S)           # launches per step in the counter run (its shards)
    steady = (per_step * av * 4.0 / 1024.0) / (step_us * 1e-6 * 2.39e9) if av and step_us > 0 else None
    if steady is not None and S > 1:
        head += f"; this run, {S} launches in flight: about {steady:.2f} busy"
    top = max(vb, steady or 0.0)
    if top >= 0.6:
        kind = ("limited by instruction issue: with every workgroup slot taken the step time is throughput, and the waves of a SIMD "
                "(vector, scalar and branch instructions in order, LDS and memory waits, barriers) do not present a vector "
                "instruction more often") if vb < 0.6 else "limited by vector-instruction issue"
    elif wa is not None and wa >= 0.4:
        kind = "limited by latency: dependent instruction chains and memory round trips of the waves, not issue or bandwidth"
    else:
        kind = "neither issue- nor bandwidth-limited: occupancy / launch ramp"
    return f"{head}: {kind}; roofline by {'FLOPs: fp32 VALU' if valu_bound else 'bytes: HBM'}"


def cpu_baseline(n_a: int, r_avoid: float, target_seconds: float = 20.0):
    """The CPU path on a bounded sample of the same workload, all host threads we may use."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import numpy as np
    import oracle_lib as ol
    from marl_amd.shapes import load_shapes
    # every core this process may run on (its affinity mask), unless OMP_NUM_THREADS names the host's CPU share
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    omp = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    threads = max(1, min(omp, usable) if omp > 0 else usable)
    kind = "port"
    if os.path.isfile(ol.REF_SO):
        try:
            ol.bind(ol.REF_SO)
            kind = "reference"
        except Exception:
            ol.bind(None)
    l_cells, grids = load_shapes()
    cfg = ol.make_cfg(n_a=n_a, r_avoid=r_avoid)
    n_g_max = (max(g.shape[0] for g in grids) + 31) // 32 * 32
    rng = np.random.default_rng(0)

    def run(E, T):
        sid, p, dp, grid, n_g, l_cell = ol.batch_reset(226, 0, 0, E, n_a, grids, l_cells, n_g_max)
        act = rng.uniform(-1, 1, (T, E, 2, n_a)).astype(np.float32)
        t0 = time.perf_counter()
        ol.run_envs(cfg, T, threads, p, dp, act, grid, n_g, l_cell)
        return time.perf_counter() - t0

    # calibrate on a small sample, then size the sample for ~target_seconds
    E0, T0 = threads * 4, 40
    run(threads, 5)                                          # thread start-up and first touches stay out of the calibration
    dt = run(E0, T0)
    per_env_step = dt * threads / (E0 * T0)                  # thread-seconds per env-step
    T = EPISODE_LENGTH
    E = max(threads, int(target_seconds * threads / (per_env_step * T)) // threads * threads)
    dt = run(E, T)
    ol.bind(None)
    return {"value": E * T * n_a / dt, "unit": "agent-steps/s", "cores": threads, "kind": kind,
            "cores_usable": usable, "cores_limit": "OMP_NUM_THREADS" if 0 < omp < usable else "affinity mask",
            "sample": f"{E} envs x {n_a} agents x {T} steps, {dt:.1f} s wall on {threads} threads "
                      f"({'compiled reference AssemblyEnv.cpp' if kind == 'reference' else 'C port of the reference'} "
                      f"driven by the C step loop of oracle/asm_oracle.c)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=400)
    ap.add_argument("--envs-per-gpu", type=int, default=4096)
    ap.add_argument("--agents", type=int, default=30)
    ap.add_argument("--graph-steps", type=int, default=0,
                    help="steps per captured HIP graph (must divide the episode length of 200; 0 = launch every step from the host)")
    ap.add_argument("--repeats", type=int, default=0,
                    help="timed windows of --steps steps each, run back to back; the reported step time is the median "
                         "window (default: at least 5, and enough for 400 timed steps in total)")
    ap.add_argument("--streams", type=int, default=0,
                    help="shards of the rank's environments, each stepped on its own HIP stream (launches in flight); "
                         "default: 3, and 4 for swarms of 256 agents or more (one workgroup per environment and CU); a "
                         "stream's next launch waits for the last workgroup of its previous one, and with three or four "
                         "launches in flight those ends overlap better than with two (a process has four hardware queues: "
                         "more streams than that share queues)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--keep-indices", nargs="?", const="both", default=None, choices=["both", "sensed"],
                    help="also materialise the sensed/occupied index lists (both) or the sensed list only")
    ap.add_argument("--periodic", action="store_true", help="periodic arena instead of walls (is_boundary=False, ENV:97-101)")
    ap.add_argument("--regime", choices=["random", "packed"], default="random",
                    help="random: the reference's reset (ENV:197-210), most agents outside the shape; packed: every "
                         "agent starts on a cell of its shape (the converged regime: full sensing, occupancy filter, sub-sampling)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from marl_amd.sharding import StreamShards, shard_range, reduce_episode_stats

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
        args.gpus = world
    # Rehearsal knobs for a one-GPU box: MARL_AMD_BENCH_ONE_DEVICE=1 puts every rank on device 0 and
    # MARL_AMD_BENCH_BACKEND=gloo replaces RCCL (which refuses two ranks on one device).  Not for measurements.
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

    trace = os.environ.get("MARL_AMD_BENCH_TRACE") == "1"

    def mark(what):
        if trace:
            print(f"[bench rank {rank}] {what}", file=sys.stderr, flush=True)

    mark("process group ready")
    if args.streams <= 0:
        args.streams = 4 if args.agents >= 256 else 3
    E, n_a, K, W, S = args.envs_per_gpu, args.agents, args.steps, args.warmup, args.streams
    # The rank's environments as S shards on S HIP streams (contiguous global ids: the same environments
    # as one batch of E); S launches are in flight, one shard's ramp and tail under the others' steady state.
    shards = StreamShards(E, n_a, parts=S, device=device, seed=226, env_id_offset=shard_range(rank, world, E)[0],
                          keep_indices={None: False, "both": True, "sensed": "sensed"}[args.keep_indices], is_boundary=not args.periodic)

    def all_reduce_stats(v):
        # with RCCL the device vector is reduced in place, asynchronously (a gloo rehearsal goes through the host)
        return reduce_episode_stats(v)

    def reset_one(k, env, episode=None):
        env.reset(episode=episode)
        if args.regime == "packed":                           # agents onto random cells of their own shape, + 2 cm jitter
            g = torch.Generator(device=device).manual_seed(1000 + 7 * env.episode + k)
            cell = (torch.rand(env.E, n_a, device=device, generator=g) * env.n_g[:, None]).long().clamp_(max=env.n_g_max - 1)
            pos = torch.gather(env.grid_center, 2, cell[:, None, :].expand(env.E, 2, n_a))
            env.p.copy_(pos + 0.02 * torch.randn(env.E, 2, n_a, device=device, generator=g))
            env.observe()

    shards.on_each(lambda k, env: reset_one(k, env, 0))
    C = args.graph_steps if args.graph_steps > 0 and EPISODE_LENGTH % args.graph_steps == 0 else 0
    # 64 precomputed action tensors per shard, used in turn (whatever --steps says: a shorter cycle is a tamer workload)
    ring = 64 if C == 0 else next(r for r in range(min(C, 64), 0, -1) if C % r == 0)

    def draw_actions(k, env):
        a = torch.empty(ring, env.E, 2, n_a, dtype=torch.float32, device=device)
        for j in range(ring):
            env.random_actions(out=a[j], step=j)
        return a

    actions = shards.on_each(draw_actions)
    n_g_mean = float(torch.cat(shards.on_each(lambda k, env: env.n_g.double())).mean())
    totals = torch.zeros(S, 8, dtype=torch.float64, device=device)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # C consecutive steps recorded into one HIP graph over the shard streams (the step allocates nothing and never
    # synchronises): one host launch per C steps.  Episode ends (statistics all-reduce, reset) stay outside.
    graph = None
    if C:
        for t in range(3):                                    # lazy kernel loading must not happen under capture
            shards.step([a[t % ring] for a in actions])
        shards.on_each(lambda k, env: reset_one(k, env, 0))
        graph = shards.capture(lambda: [shards.step([a[t % ring] for a in actions]) for t in range(C)])

    # Shard k's episodes end at steps = -phase[k] (mod 200): the shards take turns, so one shard's statistics
    # all-reduce and reset run under the other shards' step kernels instead of leaving the chip idle.  With graphs
    # every shard ends at the graph boundary.  (Shard k > 0 therefore starts with a shortened first episode.)
    phase = [0 if C else k * EPISODE_LENGTH // S for k in range(S)]

    def episode_end(k, env):                                # on shard k's stream
        mark(f"episode end, shard {k}")
        totals[k].add_(all_reduce_stats(env.episode_stats()))       # RCCL SUM of 8 doubles over xGMI
        mark(f"statistics reduced, shard {k}")
        reset_one(k, env)

    def run_steps(n, t0=0):
        t, end = t0, t0 + n
        while t < end:
            if graph is not None and t % C == 0 and end - t >= C:   # C divides the episode length: no episode end inside
                graph.replay()
                t += C
            else:
                shards.step([a[t % ring] for a in actions])
                t += 1
            for k in range(S):                              # end of shard k's episode (TRAIN:99,107,177-197)
                if (t + phase[k]) % EPISODE_LENGTH == 0:
                    shards.on_each(episode_end, only=k)

    # untimed: W warm-up steps, plus one pass through the episode-boundary path so that
    # one-time costs (lazy kernel loading, RCCL communicator set-up) stay out of the timing
    barrier()
    mark("environment reset, actions drawn")
    # The step time settles only after about 40 ms (1000 steps) of continuous load (clock ramp): when fewer warm-up
    # steps are asked for, the difference is stepped first, episode ends included, so that the timed region starts in
    # the run's steady state whatever --warmup says: the shards' episodes staggered, each somewhere inside an episode
    # (the first steps after a reset are the expensive ones -- half the environments start clustered -- and a timed
    # region that began right after a reset of every shard would measure mostly those).
    pre = max(SETTLE_STEPS - W, 0)
    run_steps(pre)
    run_steps(W, pre)
    mark("warm-up steps done")
    shards.on_each(lambda k, env: totals[k].add_(all_reduce_stats(env.episode_stats())))
    shards.synchronize()
    totals.zero_()
    mark("warm-up all-reduce done")
    barrier()
    mark("barrier passed, timing")
    # R windows of exactly K steps each, back to back (nothing synchronises between them), bracketed as a whole by
    # barrier + synchronize.  Window boundaries are HIP events on every shard stream; a window's duration is the
    # longest of its per-stream intervals, and the reported step time is the MEDIAN window / K: the first window
    # absorbs the pipeline fill after the barrier and the last one the drain, whatever K is.
    # A window shorter than an episode sees only a part of the episode's cycle (every shard resets once per
    # EPISODE_LENGTH steps, and the steps after a reset cost more): such windows are grouped into blocks of G windows
    # that span a whole episode, and the step time is the median over the blocks (at least five).
    G = 1 if K >= EPISODE_LENGTH else -(-EPISODE_LENGTH // max(K, 1))
    R = args.repeats if args.repeats > 0 else G * max(5, -(-400 // max(G * K, 1)))
    G = min(G, R)
    # (An event record costs its stream a few microseconds between two launches, so the events sit at the block boundaries:
    # after every window when a window spans an episode, after every G windows otherwise; a block's windows share its time.)
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(R // G + 1)] for _ in range(S)]
    t0 = time.perf_counter()
    for k in range(S):
        evs[k][0].record(shards.streams[k])
    for r in range(R // G * G):
        run_steps(K, pre + W + r * K)
        if (r + 1) % G == 0:
            for k in range(S):
                evs[k][(r + 1) // G].record(shards.streams[k])
    barrier()
    mark("timed region done")
    wall_all = time.perf_counter() - t0
    R = R // G * G
    win_seq = [max(evs[k][r // G].elapsed_time(evs[k][r // G + 1]) for k in range(S)) / G for r in range(R)]
    if os.environ.get("MARL_AMD_BENCH_WINDOWS") == "1" and rank == 0:
        print("windows ms:", " ".join(f"{w:.3f}" for w in win_seq), file=sys.stderr)
    win_ms = sorted(win_seq)
    blocks = sorted(sum(win_seq[b * G:(b + 1) * G]) / G for b in range(R // G))      # ms per K steps, block by block
    nb = len(blocks)
    med_ms = blocks[nb // 2] if nb % 2 else 0.5 * (blocks[nb // 2 - 1] + blocks[nb // 2])
    # per stream: launches back to back -> the launch period of the step kernel on that stream, over all windows
    ev_ms = sum(evs[k][0].elapsed_time(evs[k][R // G]) for k in range(S)) / S / R
    tmax = torch.tensor([med_ms * 1e-3, wall_all], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    wall, wall_all = float(tmax[0]), float(tmax[1])

    if rank == 0:
        agent_steps = world * E * n_a * K
        value = agent_steps / wall
        b_alg = b_alg_per_agent_step(n_a, n_g_mean)
        f_alg = f_alg_per_agent_step(n_a, n_g_mean)
        resets = sum(1 for t in range(pre + W, pre + W + R * K) for k in range(S) if (t + 1 + phase[k]) % EPISODE_LENGTH == 0)
        kern_ms = ev_ms / K                                   # launch period of one stream
        per_launch = E / S * n_a                              # agent-steps of one launch (one shard)
        per_gpu = value / world                               # agent-steps/s of one GPU
        # The roofline that bounds the step: arithmetic intensity F_alg / B_alg against the machine balance
        # (157.3 TFLOP/s / 8 TB/s = 19.7 FLOP/B).  30 agents: 10.4 FLOP/B -> HBM; 1024 agents: 34.7 -> fp32 VALU.
        valu_bound = f_alg / b_alg > VALU_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)
        env0 = shards.envs[0]
        # (observation rows are stored non-temporal on the lattice path unless the schedule setting's bit 7 says otherwise)
        nt_off = (int(os.environ.get("MARL_AMD_FAST_T", "4") or 4) & 128) != 0
        kname = step_kernel_name(env0.step_schedule(), args.periodic, {None: 0, "sensed": 1, "both": 2}[args.keep_indices],
                                 bool(env0.step_schedule() & 16) and not nt_off)
        # committed counters / traces are quoted only for the kernel they were taken on (a periodic arena, index lists or a
        # measurement knob run another instantiation) and for the reference's reset
        same = lambda t: t is not None and args.regime == "random" and any(
            k.replace("void ", "") == kname.split(" (")[0] for k in (t["kernel"] if isinstance(t["kernel"], list) else [t["kernel"]]))
        tr = measured_traffic(E, n_a, S)
        tr = tr if same(tr) else None
        pmc_step_bytes = tr["hbm_bytes_per_launch"] * tr["workload"].get("streams", S) if tr else 0.0
        kt = profiled_kernel(E, n_a, S)
        kt = kt if same(kt) and kt["workload"].get("streams") == S else None     # (a launch's duration depends on its size)
        if valu_bound:
            achieved, peak, unit = per_gpu * f_alg / 1e12, VALU_PEAK_TFLOPS, "TFLOP/s"
        else:
            achieved, peak, unit = per_gpu * b_alg / 1e9, HBM_PEAK_GBS, "GB/s"
        if n_a == 30 and E == 4096:
            which = "configs[1] of BASELINE.json"
        elif n_a == 30 and E == 8192:
            which = "one rank's shard of configs[2] of BASELINE.json (65536 envs over 8 GPUs)"
        elif n_a == 1024 and E == 256:
            which = "configs[3] of BASELINE.json (large swarm)"
        else:
            which = "not a BASELINE.json configuration"
        out = {
            "metric": f"agent-steps/sec (whole node) for batched {n_a}-agent assembly envs at 1/2/4/8 GPUs",
            "value": value, "unit": "agent-steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": wall / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "timing": {"clock": "HIP events on the shard streams at the boundaries of blocks of windows: one window when it "
                                "spans an episode, else as many as span one (an event after every short window costs the "
                                "streams 2 %); value = agent-steps of one window / (median block / windows per block), "
                                "max over ranks",
                       "windows": R, "steps_per_window": K, "windows_per_block": G,
                       "window_ms_min_median_max": [win_ms[0], med_ms, win_ms[-1]],
                       "host_wall_ms_per_step_all_windows": wall_all / (R * K) * 1e3,
                       "untimed_settle_steps": max(SETTLE_STEPS, W)},
            "config": {"workload": f"{E} parallel assembly envs x {n_a} agents per GPU, random U(-1,1) actions "
                                   f"({which}), 7 reference shapes, {'periodic arena' if args.periodic else 'walls on'}, llm_rl prior on"
                                   + (", agents started inside their shapes" if args.regime == "packed" else ""),
                       "envs_per_gpu": E, "agents": n_a, "global_envs": world * E, "mean_n_g": n_g_mean,
                       "episode_length": EPISODE_LENGTH, "episode_resets_in_timed_region": resets,
                       "state": "float32 in HBM; decisions evaluated in float64",
                       "streams_per_gpu": S, "steps_per_hip_graph": C,
                       "parallelism": f"env-sharded x{world} GPUs x{S} HIP streams, RCCL all-reduce of 8 doubles per episode"},
            # achieved = (agent-steps/s of one GPU) x algorithmic bytes (or FLOPs) per agent-step; frac = achieved / peak
            "roofline": {"bound": "valu" if valu_bound else "hbm", "achieved": achieved, "peak": peak, "unit": unit,
                         "frac": achieved / peak,
                         # PMC bytes per launch x the counter run's launches per step / this run's step time, in GB/s
                         "traffic": (pmc_step_bytes / (wall / K) / 1e9) if tr else None,
                         "traffic_bytes_per_launch": tr["hbm_bytes_per_launch"] if tr else None,
                         "traffic_launches_per_step": tr["workload"].get("streams", S) if tr else None,
                         # the same traffic against what a plain float4 copy reaches on this chip (6.29 of the 8 TB/s)
                         "traffic_frac_of_copy_rate": (pmc_step_bytes / (wall / K) / 1e9 / HBM_COPY_GBS) if tr else None,
                         "traffic_source": (tr["source"] + " (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE)") if tr else None,
                         # rocprof's VALUBusy formula on the same PMC passes (counters serialise the launches, so
                         # this is one launch alone on the chip)
                         "valu_busy_one_launch": tr.get("valu_busy") if tr else None,
                         "limiter": limiter_note(tr, valu_bound, S, wall / K * 1e6),
                         "launches_in_flight": S, "alg_bytes_per_launch": per_launch * b_alg,
                         "alg_flops_per_launch": per_launch * f_alg,
                         "kernel": kname + " (asm_batch_step_f32)",
                         # rocprofv3 --kernel-trace --stats of this workload, committed under profiles/: the kernel's own
                         # average duration; the gap is this run's launch period of a stream minus that duration
                         "kernel_profiled": kt["kernel"] if kt else None,
                         "kernel_avg_us_profiled": kt["avg_us"] if kt else None,
                         "launch_gap_us": (kern_ms * 1e3 - kt["avg_us"]) if kt else None,
                         "kernel_trace_source": kt["source"] if kt else None,
                         # (per-dispatch tracing stretches a stream's launch period, so fewer launches overlap under it than
                         # in this run: the traced duration is that of a less crowded chip)
                         "kernel_trace_note": trace_note(kt, S, wall / K * 1e3),
                         # HIP events on each shard's stream around all its timed launches, / launches: the launch period
                         # of a stream = kernel duration (rocprofv3 --kernel-trace: profiles/) + dispatch gap
                         "avg_launch_ms": kern_ms,
                         "alg_bytes_per_agent_step": b_alg, "alg_flops_per_agent_step": f_alg,
                         "agent_steps_per_launch": per_launch},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(n_a, shards.envs[0].r_avoid)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
