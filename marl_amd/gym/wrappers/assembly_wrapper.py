"""AssemblySwarmWrapper — multi-agent facade + formation metrics, with the interface of
the reference wrapper (WRAP = cus_gym/gym/wrappers/customized_envs/assembly_wrapper.py).
The metrics are vectorised numpy restatements of WRAP:48-165 (same definitions; the
reference loops cell by cell / agent by agent)."""
import numpy as np

from ..core import Wrapper


class Agent:
    def __init__(self, adversary=False):
        self.adversary = adversary


class AssemblySwarmWrapper(Wrapper):
    def __init__(self, env, args):
        super().__init__(env)
        env.__reinit__(args)                                   # WRAP:35
        self.num_agents = self.env.n_a
        self.agents = [Agent() for _ in range(self.num_agents)]
        self.agent_types = ["agent"]                           # one shared learner for the whole team
        self.action_space = self.env.action_space
        self.observation_space = self.env.observation_space
        print("Assembly environment initialized successfully.")

    def _cell_agent_dist(self):
        # d[g, j] = |p_j - cell_g|
        diff = self.p[:, None, :] - self.grid_center[:, :, None]
        return np.sqrt(diff[0] ** 2 + diff[1] ** 2)

    def coverage_rate(self):
        """Fraction of cells with an agent closer than r_avoid/2 (WRAP:48-72)."""
        return float((self._cell_agent_dist() < self.r_avoid / 2).any(axis=1).sum() / self.n_g)

    def _nearest_neighbour_dist(self):
        diff = self.p[:, None, :] - self.p[:, :, None]
        d = np.sqrt(diff[0] ** 2 + diff[1] ** 2)
        return np.where(d != 0, d, np.inf).min(axis=1)         # WRAP:93-94 drops exact zeros

    def distribution_uniformity(self):
        """(var(m) - min m) / (max m - min m) over nearest-neighbour distances m (WRAP:74-101)."""
        m = self._nearest_neighbour_dist()
        return float((np.var(m) - np.min(m)) / (np.max(m) - np.min(m)))

    def voronoi_based_uniformity(self):
        """Same normalised variance over the number of cells nearest to each agent (WRAP:103-129)."""
        owner = np.argmin(self._cell_agent_dist(), axis=1)
        cnt = np.bincount(owner, minlength=self.n_a).astype(np.float64)
        return float((np.var(cnt) - np.min(cnt)) / (np.max(cnt) - np.min(cnt)))

    def count_collisions(self):
        """Contact pairs of the state the last step started from (WRAP:131-145)."""
        return int(np.sum(np.triu(self.env.is_collide_b2b, k=1)))

    def average_distance_to_target(self):
        """Mean distance from each agent to its nearest cell (WRAP:147-165)."""
        return float(self._cell_agent_dist().min(axis=0).sum() / self.n_a)

    def get_formation_metrics(self):
        return {"coverage_rate": self.coverage_rate(),
                "distribution_uniformity": self.distribution_uniformity(),
                "voronoi_uniformity": self.voronoi_based_uniformity(),
                "num_collisions": self.count_collisions(),
                "avg_distance_to_target": self.average_distance_to_target()}
