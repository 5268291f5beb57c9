"""Rollout inference alone: the library's matrix-core policy kernel against PyTorch (autocast bf16, float32) on the
observation tensor of E environments, one JSON line.    python tools/bench_policy.py [E] [n_a] [hidden] [obs_dim]"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402
from marl_amd.policy_kernel import FusedPolicyBF16  # noqa: E402
from marl_amd.rollout import MLPNetwork  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n_a = int(sys.argv[2]) if len(sys.argv) > 2 else 30
hidden = int(sys.argv[3]) if len(sys.argv) > 3 else 256
d = int(sys.argv[4]) if len(sys.argv) > 4 else 192
net = MLPNetwork(d, 2, hidden, constrain_out=True).cuda()
fused = FusedPolicyBF16(net)
obs = torch.randn(E, d, n_a, device="cuda") * 0.5
out = torch.empty(E, 2, n_a, device="cuda")


def timed(fn, n=200):
    for _ in range(20):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


@torch.no_grad()
def torch_bf16():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out.copy_(net(obs.transpose(1, 2).reshape(E * n_a, d)).float().view(E, n_a, 2).transpose(1, 2))


@torch.no_grad()
def torch_f32():
    out.copy_(net(obs.transpose(1, 2).reshape(E * n_a, d)).view(E, n_a, 2).transpose(1, 2))


flops = 2.0 * E * n_a * (d * hidden + 2 * hidden * hidden + hidden * 2)
us = {"fused_bf16": timed(lambda: fused.forward(obs, out)), "torch_autocast_bf16": timed(torch_bf16), "torch_float32": timed(torch_f32)}
print(json.dumps({"workload": f"policy MLP {d}-{hidden}-{hidden}-{hidden}-2 for {E} x {n_a} agents", "us_per_forward": us,
                  "fused_tflops": flops / us["fused_bf16"] / 1e6, "algorithmic_gflop": flops / 1e9,
                  "obs_bytes_read_MB": E * n_a * d * 4 / 1e6}))
