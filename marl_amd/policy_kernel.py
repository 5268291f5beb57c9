"""Rollout inference of the policy network through the library's matrix-core kernel (asm_policy_mlp_forward_bf16,
include/marl_amd_env.h): packs the four Linear layers of an MLPNetwork (networks.py:6-44) into the kernel's per-lane
bf16 fragments and calls it on the environment's observation tensor.  bf16 operands, fp32 accumulation; training
still runs on the float32 PyTorch network."""
from __future__ import annotations

import torch

from . import _lib


def _pack(W: torch.Tensor, rows: int, ksteps: int, permuted: bool) -> torch.Tensor:
    """W [out][in] float32 -> [ksteps][rows / 32][64 lanes][8] bf16: lane l, element j of (k-step s, row tile m) holds
    W[32 m + (l & 31)][k], with k = 16 s + 8 (l >> 5) + j (first layer: operands in natural order) or
    k = 32 (s >> 1) + 16 (s & 1) + 8 (j >> 2) + 4 (l >> 5) + (j & 3) (the order in which a 32 x 32 accumulator tile
    presents its rows when it becomes the next product's operand).  Zero beyond W's bounds."""
    dev = W.device
    Wp = torch.zeros(rows, 16 * ksteps, dtype=torch.float32, device=dev)
    Wp[:W.shape[0], :W.shape[1]] = W
    lane = torch.arange(64, device=dev)
    r, h = lane & 31, lane >> 5
    j = torch.arange(8, device=dev)
    s = torch.arange(ksteps, device=dev)
    if permuted:
        k = (32 * (s >> 1) + 16 * (s & 1))[:, None, None] + (4 * h)[None, :, None] + (8 * (j >> 2) + (j & 3))[None, None, :]
    else:
        k = (16 * s)[:, None, None] + (8 * h)[None, :, None] + j[None, None, :]
    m = torch.arange(rows // 32, device=dev)
    row = (32 * m)[:, None, None, None] + r[None, None, :, None]                     # [M][1][64][1]
    out = Wp[row.expand(-1, ksteps, -1, 8), k[None].expand(rows // 32, -1, -1, -1)]       # [M][S][64][8]
    return out.transpose(0, 1).to(torch.bfloat16).contiguous()


class FusedPolicyBF16:
    """forward(obs (E, obs_dim, n_a) float32, out (E, act_dim, n_a) float32) for a marl_amd.rollout.MLPNetwork with
    tanh output; repack() after the network's weights changed."""

    def __init__(self, policy):
        self.lib = _lib.load()
        self.policy = policy
        self.hidden = policy.fc1.out_features
        self.obs_dim = policy.fc1.in_features
        self.act_dim = policy.fc4.out_features
        if self.hidden not in (32, 64, 128, 256) or self.act_dim > 32 or not policy.constrain_out:
            raise ValueError("the fused policy kernel takes hidden widths 32/64/128/256, at most 32 outputs and a tanh output")
        self.repack()

    @torch.no_grad()
    def repack(self) -> None:
        p, H = self.policy, self.hidden
        k1 = (self.obs_dim + 15) // 16
        self.w1 = _pack(p.fc1.weight.detach().float(), H, k1, False)
        self.w2 = _pack(p.fc2.weight.detach().float(), H, H // 16, True)
        self.w3 = _pack(p.fc3.weight.detach().float(), H, H // 16, True)
        self.w4 = _pack(p.fc4.weight.detach().float(), 32, H // 16, True)
        b4 = torch.zeros(32, dtype=torch.float32, device=self.w1.device)
        b4[:self.act_dim] = p.fc4.bias.detach().float()
        self.bias = torch.cat([p.fc1.bias.detach().float(), p.fc2.bias.detach().float(), p.fc3.bias.detach().float(), b4]).contiguous()

    def forward(self, obs: torch.Tensor, out: torch.Tensor, stream: int = None, noise_scale: float = 0.0, epsilon: float = 0.0,
                seed: int = 0, step: int = 0, env0: int = 0) -> torch.Tensor:
        """out = policy(obs); with noise_scale / epsilon > 0 the exploration of agents.py:86-94 is applied in the kernel
        (counter-based draws keyed by seed, global environment id env0 + e, agent and `step`: advance step per call)."""
        E, d, n = obs.shape
        if d != self.obs_dim or out.shape != (E, self.act_dim, n) or obs.dtype != torch.float32 or out.dtype != torch.float32 \
                or not obs.is_contiguous() or not out.is_contiguous() or obs.device.type != "cuda":
            raise ValueError("obs must be a contiguous float32 (E, obs_dim, n_a) device tensor and out (E, act_dim, n_a)")
        if stream is None:
            stream = torch.cuda.current_stream(obs.device).cuda_stream
        _lib.check(self.lib.asm_policy_mlp_act_bf16(E, n, d, self.hidden, self.act_dim, obs.data_ptr(), self.w1.data_ptr(),
                                                    self.w2.data_ptr(), self.w3.data_ptr(), self.w4.data_ptr(),
                                                    self.bias.data_ptr(), out.data_ptr(), float(noise_scale), float(epsilon),
                                                    int(seed) & (2 ** 64 - 1), int(step) & 0xFFFFFFFF, int(env0), stream),
                   "asm_policy_mlp_act_bf16")
        return out
