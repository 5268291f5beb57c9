"""The matrix-core policy kernel (asm_policy_mlp_forward_bf16) against PyTorch: the same network evaluated in float32
on bf16-rounded weights and activations (what the kernel computes, up to the order of the fp32 sums), and the plain
float32 network (the bf16 effect itself).  Exact-integer data first: a wrong fragment or row mapping cannot hide."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _reference_bf16(torch, net, x):
    """x (N, d) float32 -> the network with weights and layer inputs rounded to bf16, products and sums in float32."""
    import torch.nn.functional as F
    r = lambda t: t.detach().to(torch.bfloat16).to(torch.float32)
    h = r(x)
    for fc in (net.fc1, net.fc2, net.fc3):
        h = r(F.leaky_relu(h.double() @ r(fc.weight).double().t() + fc.bias.detach().double()).float())
    return torch.tanh((h.double() @ r(net.fc4.weight).double().t() + net.fc4.bias.detach().double()).float())


@pytest.mark.parametrize("hidden,d,n_a,E,act", [(256, 192, 30, 70, 2), (64, 40, 7, 33, 2), (128, 192, 100, 5, 3), (32, 24, 32, 9, 1),
                                                (256, 184, 1024, 2, 2)])
def test_policy_kernel_matches_pytorch(torch_cuda, hidden, d, n_a, E, act):
    torch = torch_cuda
    from marl_amd.rollout import MLPNetwork
    from marl_amd.policy_kernel import FusedPolicyBF16
    torch.manual_seed(hidden + n_a)
    net = MLPNetwork(d, act, hidden, constrain_out=True).cuda()
    # 1. small integers everywhere: every product and sum is exact in bf16 / fp32, so the result must match to the last bit
    with torch.no_grad():
        for fc in (net.fc1, net.fc2, net.fc3, net.fc4):
            fc.weight.copy_(torch.randint(-1, 2, fc.weight.shape, device="cuda").float() * (torch.rand(fc.weight.shape, device="cuda") < 0.05))
            fc.bias.copy_(torch.randint(-2, 3, fc.bias.shape, device="cuda").float())
    fused = FusedPolicyBF16(net)
    obs = torch.randint(-2, 3, (E, d, n_a), device="cuda").float()
    out = torch.full((E, act, n_a), 7.0, device="cuda")
    fused.forward(obs, out)
    x = obs.transpose(1, 2).reshape(E * n_a, d)
    want = _reference_bf16(torch, net, x).view(E, n_a, act).transpose(1, 2)
    assert torch.allclose(out, want, rtol=0, atol=2e-6), float((out - want).abs().max())     # tanh implementations differ in the last ulp
    # 2. a freshly initialised network on observation-like values
    torch.manual_seed(1)
    net = MLPNetwork(d, act, hidden, constrain_out=True).cuda()
    fused = FusedPolicyBF16(net)
    obs = torch.randn(E, d, n_a, device="cuda") * 0.5
    fused.forward(obs, out)
    x = obs.transpose(1, 2).reshape(E * n_a, d)
    want = _reference_bf16(torch, net, x).view(E, n_a, act).transpose(1, 2)
    assert float((out - want).abs().max()) < 2e-3                     # bf16 re-rounding of activations that differ in the last fp32 bit
    with torch.no_grad():
        full = net(x).view(E, n_a, act).transpose(1, 2)
    assert float((out - full).abs().max()) < 3e-2                     # the price of bf16 operands


def test_learner_uses_the_kernel_for_bf16_rollouts(torch_cuda):
    torch = torch_cuda
    from marl_amd.batched import BatchedAssemblyEnv
    from marl_amd.rollout import DDPGLearner
    env = BatchedAssemblyEnv(64, 30, seed=4)
    obs = env.reset()
    learner = DDPGLearner(env.obs_dim_agent, device="cuda", act_dtype=torch.bfloat16)
    assert learner.fused_policy is not None
    a = torch.empty(64, 2, 30, device="cuda"); b = torch.empty_like(a)
    learner.act(obs, a)
    learner.fused_policy = None                                       # PyTorch autocast path
    learner.act(obs, b)
    assert float((a - b).abs().max()) < 3e-2
    # after an update the packed weights follow the network
    learner = DDPGLearner(env.obs_dim_agent, device="cuda", act_dtype=torch.bfloat16)
    learner.act(obs, a)
    N = 256
    x = torch.randn(N, env.obs_dim_agent, device="cuda")
    learner.update(x, torch.rand(N, 2, device="cuda") * 2 - 1, torch.rand(N, 1, device="cuda"), x + 0.01, torch.zeros(N, 1, device="cuda"))
    learner.act(obs, b)
    with torch.no_grad():
        full = learner.policy(obs.transpose(1, 2).reshape(-1, env.obs_dim_agent)).view(64, 30, 2).transpose(1, 2)
    assert float((b - full).abs().max()) < 3e-2 and float((a - b).abs().max()) > 0
