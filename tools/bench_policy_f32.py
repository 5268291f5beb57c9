"""Float32 rollout inference of the reference's policy network (networks.py:22-44) for one 4096 x 30 batch through
PyTorch: time per forward under the BLAS back ends / selection modes PyTorch-ROCm offers, with and without the
observation transpose copy.  One JSON line per variant.

    python tools/bench_policy_f32.py [--envs 4096] [--agents 30] [--hidden 256]
    PYTORCH_TUNABLEOP_ENABLED=1 python tools/bench_policy_f32.py      # run-time GEMM selection (TunableOp)
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from marl_amd.rollout import MLPNetwork  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=4096)
ap.add_argument("--agents", type=int, default=30)
ap.add_argument("--hidden", type=int, default=256)
ap.add_argument("--iters", type=int, default=50)
a = ap.parse_args()
dev = torch.device("cuda")
torch.manual_seed(0)
E, n, d, H = a.envs, a.agents, 192, a.hidden
net = MLPNetwork(d, 2, H, True).to(dev)
obs = torch.randn(E, d, n, device=dev)
flops = 2.0 * E * n * (d * H + 2 * H * H + 2 * H)


def timed(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / a.iters


@torch.no_grad()
def fwd_copy():                       # maddpg.py:84: observations.t(), as a contiguous copy
    return net(obs.transpose(1, 2).reshape(E * n, d))


@torch.no_grad()
def fwd_given_rows(x=obs.transpose(1, 2).reshape(E * n, d).contiguous()):   # agent-major rows already in memory
    return net(x)


@torch.no_grad()
def fwd_inplace():                    # same chain, activations in place (no extra tensors)
    x = obs.transpose(1, 2).reshape(E * n, d)
    h = F.leaky_relu(torch.addmm(net.fc1.bias, x, net.fc1.weight.t()), inplace=True)
    h = F.leaky_relu(torch.addmm(net.fc2.bias, h, net.fc2.weight.t()), inplace=True)
    h = F.leaky_relu(torch.addmm(net.fc3.bias, h, net.fc3.weight.t()), inplace=True)
    return torch.tanh_(torch.addmm(net.fc4.bias, h, net.fc4.weight.t()))


@torch.no_grad()
def fwd_feature_major():              # H^T = W X^T per environment: batched small GEMMs on the [d][n] frames as they lie
    h = F.leaky_relu(torch.baddbmm(net.fc1.bias[None, :, None], net.fc1.weight[None].expand(E, -1, -1), obs), inplace=True)
    h = F.leaky_relu(torch.baddbmm(net.fc2.bias[None, :, None], net.fc2.weight[None].expand(E, -1, -1), h), inplace=True)
    h = F.leaky_relu(torch.baddbmm(net.fc3.bias[None, :, None], net.fc3.weight[None].expand(E, -1, -1), h), inplace=True)
    return torch.tanh_(torch.baddbmm(net.fc4.bias[None, :, None], net.fc4.weight[None].expand(E, -1, -1), h))


from marl_amd import _lib  # noqa: E402
rows_buf = torch.empty(E * n, d, device=dev)


@torch.no_grad()
def rows_lib():                       # the library's tiled transpose alone
    _lib.check(_lib.load().asm_obs_rows_f32(E, n, d, obs.data_ptr(), rows_buf.data_ptr(), torch.cuda.current_stream().cuda_stream), "rows")
    return rows_buf


@torch.no_grad()
def rows_torch():                     # PyTorch's strided copy alone
    return obs.transpose(1, 2).reshape(E * n, d)


@torch.no_grad()
def fwd_lib_rows():
    return net(rows_lib())


for name, fn in (("agent rows alone: library tiled transpose", rows_lib), ("agent rows alone: PyTorch strided copy", rows_torch)):
    t = timed(fn)
    print(json.dumps({"variant": name, "us": t * 1e6, "GB_per_s": 2 * obs.numel() * 4 / t / 1e9}), flush=True)
libs = ["default"]
if hasattr(torch.backends.cuda, "preferred_blas_library"):
    libs += ["hipblaslt", "cublas"]
ref = fwd_copy().clone()
for lib in libs:
    if lib != "default":
        try:
            torch.backends.cuda.preferred_blas_library(lib)
        except Exception as ex:                                     # noqa: BLE001
            print(json.dumps({"blas": lib, "error": str(ex)})); continue
    for name, fn in (("transpose copy + nn.Module", fwd_copy), ("rows given + nn.Module", fwd_given_rows), ("library transpose + nn.Module", fwd_lib_rows),
                     ("transpose copy + addmm, in-place activations", fwd_inplace), ("feature-major batched GEMMs", fwd_feature_major)):
        try:
            out = fn()
            got = out if out.dim() == 2 else out.transpose(1, 2).reshape(E * n, 2)
            err = float((got - ref).abs().max())
            t = timed(fn)
            print(json.dumps({"blas": lib, "tunableop": os.environ.get("PYTORCH_TUNABLEOP_ENABLED", "0"), "variant": name,
                              "us_per_forward": t * 1e6, "tflops": flops / t / 1e12, "max_abs_diff_vs_first": err}), flush=True)
        except Exception as ex:                                     # noqa: BLE001
            print(json.dumps({"blas": lib, "variant": name, "error": str(ex)[:200]}), flush=True)
