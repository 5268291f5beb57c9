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
b = buf.cpu().numpy(); b = b[b[:, 9] > 0]
env.lib.asm_diag_set_buffer(None)
hw = b[:, 20:24]
simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; wave = hw & 15
print("workgroups", len(b))
for s in range(4):
    print(f"wave {s}: SIMD histogram {np.bincount(simd[:, s], minlength=4).tolist()}  wave-slot histogram {np.bincount(wave[:, s], minlength=16).tolist()}")
print("workgroups whose 4 waves sit on 4 different SIMDs:", int((np.sort(simd, axis=1) == np.arange(4)).all(axis=1).sum()))
print("first 6 workgroups (simd of waves 0..3):", simd[:6].tolist())
