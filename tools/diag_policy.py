"""Phase cycles of the policy kernel's workgroups (diagnostic build, in-kernel cycle stamps by the first thread):
    python tools/diag_policy.py [E] [n_a] [hidden]"""
import ctypes as C
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault("MARL_AMD_LIB", os.path.join(REPO, "marl_amd", "csrc", "build", "libAssemblyEnv_diag.so"))
import torch  # noqa: E402
from marl_amd import _lib  # noqa: E402
from marl_amd.policy_kernel import FusedPolicyBF16  # noqa: E402
from marl_amd.rollout import MLPNetwork  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n_a = int(sys.argv[2]) if len(sys.argv) > 2 else 30
hidden = int(sys.argv[3]) if len(sys.argv) > 3 else 256
net = MLPNetwork(192, 2, hidden, constrain_out=True).cuda()
fused = FusedPolicyBF16(net)
obs = torch.randn(E, 192, n_a, device="cuda") * 0.5
out = torch.empty(E, 2, n_a, device="cuda")
for _ in range(10):
    fused.forward(obs, out)
nwg = (E * ((n_a + 31) // 32) + 7) // 8
buf = torch.zeros(nwg, 32, dtype=torch.int64, device="cuda")
lib = _lib.load()
assert lib.asm_diag_set_buffer(C.c_void_p(buf.data_ptr())) == 0
acc = []
for _ in range(10):
    buf.zero_()
    fused.forward(obs, out); torch.cuda.synchronize()
    b = buf.cpu().numpy()
    acc.append(np.diff(b[:, :18], axis=1).astype(np.float64))
    wall = (b[:, 31] - b[:, 30]) / 100.0
    start = (b[:, 30] - b[:, 30].min()) / 100.0
lib.asm_diag_set_buffer(None)
d = np.mean(acc, axis=0)
names = ["observation rows, w1, biases issued; barrier 1", "layer 1 (k-steps of 16 features, rows streamed)", "barrier 2", "copy w2a issued, activation 1",
         "barrier 3", "layer 2 first half", "barrier 4", "layer 2 second half", "activation 2", "barrier 5", "layer 3 first half", "barrier 6",
         "layer 3 second half", "activation 3", "barrier 7", "output layer", "tanh, exploration, stores"]
print(f"workgroups {nwg}; lifetime us p50/max {np.percentile(wall, 50):.1f}/{wall.max():.1f}; start us p50/max {np.percentile(start, 50):.1f}/{start.max():.1f}; "
      f"cycles per workgroup {d.sum(1).mean():.0f}")
for k, n in enumerate(names):
    print(f"  {n:55s} {d[:, k].mean():9.0f}  {100 * d[:, k].mean() / d.sum(1).mean():5.1f} %")
