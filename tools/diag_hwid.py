"""Which SIMD each wave of a k_step_fast workgroup runs on (diagnostic build, HW_ID register).
    MARL_AMD_LIB=marl_amd/csrc/build/libAssemblyEnv_diag.so python tools/diag_hwid.py"""
import ctypes as C, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault("MARL_AMD_LIB", os.path.join(REPO, "marl_amd", "csrc", "build", "libAssemblyEnv_diag.so"))
import torch
from marl_amd.batched import BatchedAssemblyEnv
env = BatchedAssemblyEnv(2048, 30)
env.reset()
for t in range(20):
    env.step(env.random_actions())
buf = torch.zeros(8192, 32, dtype=torch.int64, device="cuda")
assert env.lib.asm_diag_set_buffer(C.c_void_p(buf.data_ptr())) == 0
env.step(env.random_actions()); torch.cuda.synchronize()
b = buf.cpu().numpy(); wg = np.nonzero(b[:, 9] > 0)[0]; b = b[b[:, 9] > 0]
env.lib.asm_diag_set_buffer(None)
hw = b[:, 20:24]
simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; wave = hw & 15
print("workgroups", len(b))
for s in range(4):
    print(f"wave {s}: SIMD histogram {np.bincount(simd[:, s], minlength=4).tolist()}  wave-slot histogram {np.bincount(wave[:, s], minlength=16).tolist()}")
print("workgroups whose 4 waves sit on 4 different SIMDs:", int((np.sort(simd, axis=1) == np.arange(4)).all(axis=1).sum()))
print("first 6 workgroups (simd of waves 0..3):", simd[:6].tolist())
# per CU: do the first waves (the heaviest role) of the workgroups resident together sit on different SIMDs?
xcc = b[:, 24] & 15
se = (hw[:, 0] >> 13) & 7; sh = (hw[:, 0] >> 12) & 1
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu[:, 0]
dist = {}
for k in np.unique(key):
    s0 = np.sort(simd[key == k, 0]).tolist()
    dist[tuple(s0)] = dist.get(tuple(s0), 0) + 1
print("CUs seen:", len(np.unique(key)), "; SIMDs of the first waves of a CU's workgroups -> number of CUs:", sorted(dist.items(), key=lambda kv: -kv[1])[:8])
# workgroup index -> XCD: round-robin?
print("workgroups on XCD (workgroup index % 8):", int(((b[:, 24] & 15) == wg % 8).sum()), "of", len(b),
      "; XCD ids of workgroups 0..15:", (b[:16, 24] & 15).tolist())
