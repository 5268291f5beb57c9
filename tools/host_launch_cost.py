"""Host cost of one asm_batch_step_f32 call through the Python binding: steps of a batch so small that the GPU is
never the limit (the queue stays short), timed on the host without synchronising, with and without the per-thread
cache of derived parameters (MARL_AMD_NO_PARAM_CACHE=1 re-derives the exact thresholds on every call, as the
library did before the cache).  Then the one-stream step time of the 4096-environment benchmark batch next to it.

    python tools/host_launch_cost.py
"""
import json
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from marl_amd.batched import BatchedAssemblyEnv
    out = {}
    for E in (64, 4096):
        env = BatchedAssemblyEnv(E, 30)
        env.reset()
        a = env.random_actions()
        for _ in range(300):
            env.step(a)
        torch.cuda.synchronize()
        n = 3000
        t0 = time.perf_counter()
        for _ in range(n):
            env.step(a)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        out[f"E{E}_host_us_per_call"] = (t1 - t0) / n * 1e6
        out[f"E{E}_us_per_step_synchronised"] = (t2 - t0) / n * 1e6
    print(json.dumps(out))
else:
    res = {}
    for tag, env in (("cached", {}), ("derived_every_call", {"MARL_AMD_NO_PARAM_CACHE": "1"})):
        o = subprocess.check_output([sys.executable, os.path.abspath(__file__), "child"], env={**os.environ, **env}, text=True)
        res[tag] = json.loads(o.strip().splitlines()[-1])
    print(json.dumps(res, indent=1))
