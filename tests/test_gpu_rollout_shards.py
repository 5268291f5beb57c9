"""run_episode on a StreamShards environment (two shards acting and stepping on streams of their own) against the same
episode on one BatchedAssemblyEnv of all the environments: the replay frames must be identical, because the shards own
the same global environment ids and the policy kernel's exploration is keyed by (seed, global id, agent, time step)."""
import pytest

pytestmark = pytest.mark.gpu


def test_sharded_rollout_equals_the_single_environment(torch_cuda):
    torch = torch_cuda
    from marl_amd.batched import BatchedAssemblyEnv
    from marl_amd.replay import DeviceReplay
    from marl_amd.rollout import DDPGLearner, run_episode
    from marl_amd.sharding import StreamShards
    E, n_a, T = 66, 30, 7
    rings = []
    for sharded in (False, True):
        torch.manual_seed(3)
        env = StreamShards(E, n_a, parts=2, env_id_offset=1000) if sharded else BatchedAssemblyEnv(E, n_a, env_id_offset=1000)
        learner = DDPGLearner(env.obs_dim_agent, hidden_dim=128, act_dtype=torch.bfloat16, explore_seed=11)
        assert learner.fused_policy is not None
        rp = DeviceReplay(env, T, 2, seed=5)
        stats = [run_episode(env, learner, rp, T, noise_scale=0.4, epsilon=0.3) for _ in range(2)]
        torch.cuda.synchronize()
        assert rp.filled_slots == 2 and len(rp) == 2 * T * E * n_a
        rings.append((rp.obs_ring.clone(), rp.act_ring.clone(), rp.rew_ring.clone(), rp.prior_ring.clone(), torch.stack(stats)))
        obs, act, rew, nxt, done, pri = rp.sample(512)              # the sampler reads the frames the shards wrote
        assert torch.isfinite(obs).all() and act.abs().max() <= 1.0
    for a, b in zip(rings[0][:4], rings[1][:4]):
        assert a.abs().sum() > 0
        assert torch.equal(a, b)
    # the episode statistics are float64 sums over the environments, taken per shard and then added
    assert torch.allclose(rings[0][4], rings[1][4], rtol=1e-12, atol=0)
