"""The train phase of data-parallel learner replicas with its update recorded into HIP graphs (StagedUpdate /
GraphedTrainPhase): three graphs per update, the two flattened gradient all-reduces issued eagerly between their
replays.  Two ranks share the one GPU of the box and exchange through gloo (RCCL wants one device per rank): what
is checked is that graph replays + eager exchanges give the weights of the plain data-parallel update
(maddpg.py:99-190 with parallel=True, :138-139, :179-180) and that the replicas stay identical."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBS, ACT, HID, N, UPDATES = 192, 2, 64, 256, 6


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, REPO)
    import torch
    import torch.distributed as dist
    from marl_amd.rollout import DDPGLearner, StagedUpdate
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(50 + rank)           # every rank its own batches
    obs = torch.randn(UPDATES, N, OBS, device=dev, generator=g)
    acs = torch.rand(UPDATES, N, ACT, device=dev, generator=g) * 2 - 1
    rew = (torch.rand(UPDATES, N, 1, device=dev, generator=g) < 0.3).float()
    nxt = obs + 0.05 * torch.randn(UPDATES, N, OBS, device=dev, generator=g)
    done = torch.zeros(UPDATES, N, 1, device=dev)
    pri = torch.rand(UPDATES, N, ACT, device=dev, generator=g) * 2 - 1
    pri[:, ::3] = 0.0

    def weights(L):
        return torch.cat([p.detach().reshape(-1) for net in (L.policy, L.critic, L.target_policy, L.target_critic)
                          for p in net.parameters()]).cpu().numpy()

    res = {}
    for mode in ("plain", "graphs"):
        torch.manual_seed(7 + rank)
        L = DDPGLearner(OBS, ACT, hidden_dim=HID, device=dev, data_parallel=True, capturable=True)
        assert L.data_parallel
        if mode == "plain":
            for u in range(UPDATES):
                L.update(obs[u], acs[u], rew[u], nxt[u], done[u], acs_prior=pri[u], alpha=0.7, sync_losses=False)
                L.update_all_targets()
        else:
            k = torch.zeros(1, dtype=torch.int64, device=dev)        # which batch: a device word the last piece increments
            pick = lambda t: torch.index_select(t, 0, k)[0]
            st = StagedUpdate(L, lambda: (pick(obs), pick(acs), pick(rew), pick(nxt), pick(done), pick(pri)), lambda: 0.7,
                              dev, use_graphs=True, after_fn=lambda: k.add_(1))
            warm = st.capture(2)
            assert len(st.graphs) == 3 and warm.shape == (2, 2)
            losses = []
            for _ in range(UPDATES - 2):
                st.run_once()
                losses.append(st.losses.clone())
            assert st.launches_per_update == 5
            torch.cuda.synchronize()
            assert int(k) == UPDATES and L.niter == UPDATES
            assert all(bool(torch.isfinite(x).all()) for x in losses)
        torch.cuda.synchronize()
        res[mode] = weights(L)
    out.put((rank, res["plain"], res["graphs"]))
    dist.barrier()
    dist.destroy_process_group()


def test_graphed_data_parallel_update_matches_the_plain_one(torch_cuda):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {r: (a, b) for r, a, b in (q.get(timeout=240) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(got[0][1], got[1][1]), "replicas diverged under graph replay"
    for r in range(world):
        # same kernels, same order; the recorded pieces may pick other GEMM work-space layouts: float32 rounding at most
        np.testing.assert_allclose(got[r][1], got[r][0], rtol=1e-5, atol=1e-6)
