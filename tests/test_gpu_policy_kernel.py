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


# The first layer takes its operands through LDS slabs where an environment's rows are one aligned block (at most 32
# agents, obs_dim x n_a a multiple of 4, hidden width >= 128) and straight from memory otherwise: both kinds below, with
# 1 or 2 copy instructions per slab (n_a <= 16 or more), 3 or 4 k-steps of slots (n_a = 32 leaves room for 3), obs_dim
# not a multiple of 16 (a short last slab), fewer k-steps than slots, a rollout batch that ends inside a workgroup.
@pytest.mark.parametrize("hidden,d,n_a,E,act", [(256, 192, 30, 70, 2), (64, 40, 7, 33, 2), (128, 192, 100, 5, 3), (32, 24, 32, 9, 1),
                                                (256, 184, 1024, 2, 2), (256, 184, 32, 9, 2), (128, 200, 12, 21, 2),
                                                (256, 24, 16, 5, 2), (128, 35, 28, 11, 4), (128, 33, 30, 3, 2)])
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
    # 3. the same observations at an address that is not a multiple of 16 bytes (no slabs then): the same actions
    shifted = torch.empty(obs.numel() + 1, device="cuda")[1:].view_as(obs)
    shifted.copy_(obs)
    out2 = torch.full_like(out, 7.0)
    fused.forward(shifted, out2)
    assert float((out2 - out).abs().max()) < 2e-3                     # (the two paths sum in the same order: normally 0)


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
    keep, learner.fused_policy = learner.fused_policy, None           # PyTorch autocast path
    learner.act(obs, b)
    assert float((a - b).abs().max()) < 3e-2
    learner.fused_policy = keep
    learner.act(obs, b, noise_scale=0.5, epsilon=0.1)                 # exploration drawn by the kernel: a new draw per call
    c = torch.empty_like(b)
    learner.act(obs, c, noise_scale=0.5, epsilon=0.1)
    assert not torch.equal(b, c) and float(b.abs().max()) <= 1.0
    g = torch.Generator(device="cuda").manual_seed(1)                 # with a generator: the torch draws (seedable by the caller)
    learner.act(obs, b, noise_scale=0.5, epsilon=0.1, generator=g)
    g.manual_seed(1)
    learner.act(obs, c, noise_scale=0.5, epsilon=0.1, generator=g)
    assert torch.equal(b, c)
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


def test_policy_kernel_exploration(torch_cuda):
    """agents.py:86-94 in the kernel's epilogue: the epsilon coin is per environment, uniform actions fill [-1, 1), the
    noise is N(0, scale^2) and clamped, draws depend on (seed, step, global environment id) only."""
    torch = torch_cuda
    from marl_amd.rollout import MLPNetwork
    from marl_amd.policy_kernel import FusedPolicyBF16
    torch.manual_seed(3)
    E, n_a, d = 4096, 30, 192
    net = MLPNetwork(d, 2, 256, constrain_out=True).cuda()
    with torch.no_grad():
        net.fc4.weight.mul_(0.2)                                      # actions well inside (-1, 1)
    fused = FusedPolicyBF16(net)
    obs = torch.randn(E, d, n_a, device="cuda") * 0.5
    det = torch.empty(E, 2, n_a, device="cuda"); a = torch.empty_like(det); b = torch.empty_like(det)
    fused.forward(obs, det)
    # everybody uniform
    fused.forward(obs, a, epsilon=1.0, seed=5, step=1)
    assert float(a.min()) >= -1.0 and float(a.max()) < 1.0
    assert abs(float(a.mean())) < 5e-3 and abs(float(a.var()) - 1.0 / 3.0) < 5e-3
    # Gaussian noise, clamped
    fused.forward(obs, a, noise_scale=0.3, seed=5, step=2)
    z = (a - det) / 0.3
    inside = a.abs() < 1.0
    assert float(inside.float().mean()) > 0.95 and float(a.abs().max()) <= 1.0
    assert abs(float(z[inside].mean())) < 5e-3 and abs(float(z[inside].std()) - 1.0) < 2e-2
    assert abs(float((z[inside] ** 4).mean()) - 3.0) < 0.15            # a Gaussian's fourth moment (mild truncation by the clamp)
    zz = z.view(E, 2, n_a)
    assert abs(float((zz[:, 0] * zz[:, 1]).mean())) < 5e-3             # the two outputs of an agent are independent
    # the coin is per environment
    fused.forward(obs, a, epsilon=0.25, seed=5, step=3)
    same = (a == det).view(E, -1)
    per_env = same.all(dim=1)
    assert bool((per_env | ~same.any(dim=1)).all())                    # an environment is either untouched or all random
    assert abs(float((~per_env).float().mean()) - 0.25) < 0.03
    # draws: functions of (seed, step, global environment id)
    fused.forward(obs, b, epsilon=0.25, seed=5, step=3)
    assert torch.equal(a, b)
    fused.forward(obs, b, epsilon=0.25, seed=5, step=4)
    assert not torch.equal(a, b)
    fused.forward(obs, b, epsilon=0.25, seed=6, step=3)
    assert not torch.equal(a, b)
    fused.forward(obs, a, noise_scale=0.3, epsilon=0.1, seed=9, step=7)
    half = torch.empty(E // 2, 2, n_a, device="cuda")
    fused.forward(obs[E // 2:].contiguous(), half, noise_scale=0.3, epsilon=0.1, seed=9, step=7, env0=E // 2)
    assert torch.equal(half, a[E // 2:])


def test_fused_policy_follows_graphed_updates_and_loaded_weights(torch_cuda):
    """The packed bf16 weights must follow the float32 network whoever changes it: an optimiser step from Python, a
    train phase replayed from a HIP graph (Python sees nothing run), load_state_dict."""
    torch = torch_cuda
    from marl_amd.batched import BatchedAssemblyEnv
    from marl_amd.replay import DeviceReplay
    from marl_amd.rollout import DDPGLearner, GraphedTrainPhase, run_episode
    env = BatchedAssemblyEnv(64, 30, seed=2)
    rp = DeviceReplay(env, 8, 2, seed=0)
    learner = DDPGLearner(env.obs_dim_agent, device="cuda", act_dtype=torch.bfloat16, capturable=True, hidden_dim=64)
    run_episode(env, learner, rp, 8)
    obs = env.reset()
    a = torch.empty(64, 2, 30, device="cuda"); b = torch.empty_like(a)

    def close_to_network():
        learner.act(obs, b)
        with torch.no_grad():
            full = learner.policy(obs.transpose(1, 2).reshape(-1, env.obs_dim_agent)).view(64, 30, 2).transpose(1, 2)
        return float((b - full).abs().max()) < 3e-2

    learner.act(obs, a)
    gt = GraphedTrainPhase(env, learner, rp, batch_size=256)
    gt.run(num_updates=12)                                                # 3 warm-up updates from Python, 9 replays
    assert close_to_network() and not torch.equal(a, b)
    a.copy_(b)
    gt.run(num_updates=10)                                                # replays only
    assert close_to_network() and not torch.equal(a, b)
    other = DDPGLearner(env.obs_dim_agent, device="cuda", hidden_dim=64)
    learner.policy.load_state_dict(other.policy.state_dict())
    a.copy_(b)
    assert close_to_network() and not torch.equal(a, b)
