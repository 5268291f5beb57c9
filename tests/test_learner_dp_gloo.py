"""Data-parallel learner replicas on CPU (gloo, world size 2): rank 0's initial weights reach every rank, each
update all-reduces one flattened gradient per network, and ranks that saw DIFFERENT batches end with identical
weights, equal (to float32 rounding) to one process trained on the two batches joined — the averaged gradient
of two equal-sized batches is the gradient of the mean loss over their union."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBS, ACT, HID, N, UPDATES = 24, 2, 16, 64, 4


def _batch(seed, n=N):
    g = torch.Generator().manual_seed(seed)
    obs = torch.randn(n, OBS, generator=g)
    return (obs, torch.rand(n, ACT, generator=g) * 2 - 1, (torch.rand(n, 1, generator=g) < 0.3).float(),
            obs + 0.05 * torch.randn(n, OBS, generator=g), torch.zeros(n, 1))


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, REPO)
    from marl_amd.rollout import DDPGLearner
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    torch.manual_seed(100 + rank)                                     # replicas start DIFFERENT until the broadcast
    L = DDPGLearner(OBS, ACT, hidden_dim=HID, device="cpu", data_parallel=True)
    assert L.data_parallel
    calls = []
    real = dist.all_reduce
    dist.all_reduce = lambda t, *a, **k: (calls.append(t.numel()), real(t, *a, **k))[1]
    for u in range(UPDATES):
        L.update(*_batch(10 * u + rank), acs_prior=None)
        L.update_all_targets()
    dist.all_reduce = real
    n_cri = sum(p.numel() for p in L.critic.parameters()); n_pol = sum(p.numel() for p in L.policy.parameters())
    assert calls == [n_cri, n_pol] * UPDATES, "one flattened all-reduce per optimiser step"
    flat = torch.cat([p.detach().reshape(-1) for net in (L.policy, L.critic, L.target_policy, L.target_critic)
                      for p in net.parameters()])
    out.put((rank, flat.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def _prior(seed, n=N):
    g = torch.Generator().manual_seed(1000 + seed)
    pr = torch.rand(n, ACT, generator=g) * 2 - 1
    pr[::3] = 0.0                                                     # rows without a prior (maddpg.py:162)
    return pr


def _weights(L):
    return torch.cat([p.detach().reshape(-1) for net in (L.policy, L.critic, L.target_policy, L.target_critic)
                      for p in net.parameters()])


def _staged_worker(rank, world, port, out):
    """The update cut at its two gradient exchanges (StagedUpdate, the form GraphedTrainPhase records into HIP graphs)
    against the plain data-parallel update, same ranks, same batches."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, REPO)
    from marl_amd.rollout import DDPGLearner, StagedUpdate
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    res = {}
    for mode in ("plain", "staged"):
        torch.manual_seed(100 + rank)
        L = DDPGLearner(OBS, ACT, hidden_dim=HID, device="cpu", data_parallel=True)
        calls = []
        real = dist.all_reduce
        dist.all_reduce = lambda t, *a, **k: (calls.append(t.numel()), real(t, *a, **k))[1]
        if mode == "plain":
            for u in range(UPDATES):
                L.update(*_batch(10 * u + rank), acs_prior=_prior(10 * u + rank), alpha=0.7)
                L.update_all_targets()
        else:
            step = [0]
            after = []
            st = StagedUpdate(L, lambda: _batch(10 * step[0] + rank) + (_prior(10 * step[0] + rank),), lambda: 0.7, "cpu",
                              use_graphs=False, after_fn=lambda: (after.append(step[0]), step.__setitem__(0, step[0] + 1)))
            warm = st.capture(2)                                      # two warm-up updates, counted
            assert warm.shape == (2, 2) and bool(torch.isfinite(warm).all())
            for _ in range(UPDATES - 2):
                st.run_once()
            assert after == list(range(UPDATES)) and L.niter == UPDATES
        dist.all_reduce = real
        n_cri = sum(p.numel() for p in L.critic.parameters()); n_pol = sum(p.numel() for p in L.policy.parameters())
        assert calls == [n_cri, n_pol] * UPDATES, (mode, "one flattened all-reduce per optimiser step")
        res[mode] = _weights(L).numpy()
    out.put((rank, res["plain"], res["staged"]))
    dist.barrier()
    dist.destroy_process_group()


def test_staged_update_equals_the_plain_data_parallel_update():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_staged_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {r: (a, b) for r, a, b in (q.get(timeout=180) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        assert np.array_equal(got[r][0], got[r][1]), "the staged update must be the plain update, bit for bit"
    assert np.array_equal(got[0][0], got[1][0]), "replicas diverged"


def test_two_ranks_with_different_batches_keep_identical_weights():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(got[0], got[1]), "replicas diverged"
    # one process, rank 0's initial weights, the two batches joined
    sys.path.insert(0, REPO)
    from marl_amd.rollout import DDPGLearner
    torch.set_num_threads(1)
    torch.manual_seed(100)
    L = DDPGLearner(OBS, ACT, hidden_dim=HID, device="cpu")
    for u in range(UPDATES):
        parts = [_batch(10 * u + r) for r in range(world)]
        L.update(*[torch.cat(xs) for xs in zip(*parts)], acs_prior=None)
        L.update_all_targets()
    single = torch.cat([p.detach().reshape(-1) for net in (L.policy, L.critic, L.target_policy, L.target_critic)
                        for p in net.parameters()]).numpy()
    np.testing.assert_allclose(got[0], single, rtol=2e-5, atol=2e-6)
