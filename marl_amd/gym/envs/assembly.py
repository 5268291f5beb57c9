"""AssemblySwarmEnv — single-environment, float64, numpy-facing environment with the
interface of the reference's class of the same name
(ENV = cus_gym/gym/envs/customized_envs/assembly.py), for callers written against it:
`__reinit__(args)`, `reset() -> obs`, `step(a) -> (obs, rew, done, info, a_prior)` and
the public state attributes the trainer, evaluator and wrapper read.

Like the reference it keeps the state in numpy and calls five C symbols through
ctypes (ENV:229, 352, 455, 490, 608).  Here those symbols are the ones of
libAssemblyEnv.so built from marl_amd/csrc: each call runs the double-precision HIP
kernels on the GPU.  There is no CPU implementation behind this class.

For throughput use marl_amd.batched.BatchedAssemblyEnv (thousands of environments
per launch, state resident in HBM); this class exists so that code written for the
reference keeps working and so that parity can be checked call by call.
"""
from __future__ import annotations

import ctypes
import pickle

import numpy as np

from ... import _lib
from ...shapes import results_dict
from .. import spaces
from ..core import Env


def _dbl(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _i32(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))


def _bool(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_bool))


class AssemblySwarmEnv(Env):
    metadata = {"render.modes": ["human", "rgb_array"]}

    def __init__(self):
        # constants of ENV:17-89
        self.reward_sharing_mode = "individual"
        self.penalize_entering = True
        self.penalize_interaction = True
        self.penalize_exploration = True
        self.dim = 2
        self.n_a, self.n_o = 10, 0
        self.topo_nei_max = 6
        self.act_dim_agent = self.dim
        self.m_a, self.m_o = 1, 10
        self.size_a, self.size_o = 0.035, 0.2
        self.d_sen, self.r_avoid = 3, 0.15
        self.Vel_max, self.Vel_min, self.Acc_max = 0.8, 0.0, 1
        self.boundary_width_half = 2.4
        self.boundary_height_half = 2.4
        self.bound_center = np.zeros(2)
        self.k_ball, self.k_wall, self.c_wall, self.c_aero = 30, 100, 5, 1.2
        self.simulation_time, self.dt, self.n_frames, self.sensitivity = 0, 0.1, 1, 1
        self.figure_handle = None
        self.plot_initialized = 0
        self._lib = None

    # ------------------------------------------------------------------ set-up
    def __reinit__(self, args):
        """ENV:91-149."""
        self._lib = _lib.load()                       # fails loudly if the HIP library is missing
        self.n_a = args.n_a
        self.render_traj = getattr(args, "render_traj", False)
        self.traj_len = getattr(args, "traj_len", 15)
        self.is_collected = args.is_collected
        self.is_boundary = args.is_boundary
        self.is_periodic = not self.is_boundary
        self.dynamics_mode = args.dynamics_mode
        self.agent_strategy = args.agent_strategy
        self.is_con_self_state = args.is_con_self_state
        self.is_feature_norm = args.is_feature_norm
        self.training_method = args.training_method
        self.alpha = 1
        self.results_file = getattr(args, "results_file", None)
        if self.results_file is None:
            loaded = results_dict()                   # packaged grids of the reference's seven shapes
        else:
            with open(self.results_file, "rb") as f:  # the caller's own results.pkl (ENV:110-112)
                loaded = pickle.load(f)
        self.l_cells = loaded["l_cell"]
        self.grid_center_origins = loaded["grid_coords"]
        self.binary_images = loaded.get("binary_image")
        self.shape_bound_points_origins = loaded.get("shape_bound_points")
        self.num_train_shape = len(self.l_cells)
        self.n_gs = [g.shape[0] for g in self.grid_center_origins]
        self.r_avoid = round(np.sqrt(4 * np.min(self.n_gs) / (self.n_a * np.pi)) * np.min(self.l_cells), 2)  # ENV:122

        self.num_obs_grid_max = 80
        self.num_occupied_grid_max = 200
        self.in_flags = np.zeros(self.n_a, dtype=np.int32)
        self.sensed_index = -np.ones((self.n_a, self.num_obs_grid_max), dtype=np.int32)
        self.occupied_index = -np.ones((self.n_a, self.num_occupied_grid_max), dtype=np.int32)
        self.neighbor_index = -np.ones((self.n_a, self.topo_nei_max), dtype=np.int32)
        self.is_collide_b2b = np.random.choice([True, False], size=(self.n_a, self.n_a))   # ENV:131
        self.is_collide_b2w = np.zeros((4, self.n_a), dtype=bool)
        self.d_b2w = np.ones((4, self.n_a))

        self_flag = 1 if self.is_con_self_state else 0
        self.obs_dim_agent = 2 * self.dim * (self.topo_nei_max + 1 + self_flag) + self.dim * self.num_obs_grid_max  # ENV:806
        self.observation_space = spaces.Box(-np.inf, np.inf, shape=(self.obs_dim_agent, self.n_a), dtype=np.float32)
        self.action_space = spaces.Box(-np.inf, np.inf, shape=(self.act_dim_agent, self.n_a), dtype=np.float32)
        self.m = np.full(self.n_a, self.m_a)
        self.size = np.full(self.n_a, self.size_a, dtype=np.float64)
        self.sizes = self.size[:, None] + self.size[None, :]                                # ENV:787-793
        np.fill_diagonal(self.sizes, 0)
        if self.dynamics_mode == "Cartesian":
            self.is_Cartesian = True
        else:
            print("Wrong in linAcc_p_min")                                                  # ENV:144-145
        self.shape_frequency = np.zeros_like(self.l_cells)

    # ------------------------------------------------------------------ reset
    def reset(self):
        """ENV:151-218, same draws in the same order from numpy's global generator."""
        self.simulation_time = 0
        k = np.random.randint(0, self.num_train_shape)
        self.shape_frequency[k] += 1
        self.l_cell = self.l_cells[k]
        origin = self.grid_center_origins[k].T
        angle = np.pi * np.random.uniform(-1, 1)
        rot = np.array([[np.cos(angle), np.sin(angle)], [-np.sin(angle), np.cos(angle)]])
        self.grid_center_origin = np.dot(rot, origin)
        self.n_g = self.grid_center_origin.shape[1]
        np.random.uniform(-1.2, 1.2, (2, 1))                                                # ENV:177, value unused
        W, H = self.boundary_width_half, self.boundary_height_half
        offset = np.array([[np.random.uniform(-W + 1, W - 1), np.random.uniform(-H + 1, H - 1)]]).T
        self.grid_center = self.grid_center_origin.copy() + offset
        self.bound_center = np.array([0, 0])
        self.boundary_pos = np.array([self.bound_center[0] - W, self.bound_center[1] + H,
                                      self.bound_center[0] + W, self.bound_center[1] - H], dtype=np.float64)
        self.d_sen = 0.4
        if np.random.uniform(-1, 1) > 0:
            self.p = np.concatenate((np.random.uniform(-W, W, (1, self.n_a)), np.random.uniform(-H, H, (1, self.n_a))), axis=0)
        else:
            self.p = np.random.uniform(-1, 1, (2, self.n_a)) + \
                np.array([[np.random.uniform(-W + 1, W - 1), np.random.uniform(-H + 1, H - 1)]]).T
        self.dp = np.random.uniform(-0.5, 0.5, (self.dim, self.n_a))
        self.ddp = np.zeros((2, self.n_a))
        self.heading = np.zeros((self.dim, self.n_a))
        return self._get_obs()

    # ------------------------------------------------------------------ the five calls
    def _get_obs(self):
        """ENV:220-250 -> _get_observation."""
        self.obs = np.zeros(self.observation_space.shape)
        self.neighbor_index = -np.ones((self.n_a, self.topo_nei_max), dtype=np.int32)
        self.in_flags = np.zeros(self.n_a, dtype=np.int32)
        self.sensed_index = -np.ones((self.n_a, self.num_obs_grid_max), dtype=np.int32)
        self.occupied_index = -np.ones((self.n_a, self.num_occupied_grid_max), dtype=np.int32)
        cond = np.array([self.is_periodic, self.is_Cartesian, self.is_con_self_state, self.is_feature_norm])
        self.p = np.ascontiguousarray(self.p, dtype=np.float64)
        self.dp = np.ascontiguousarray(self.dp, dtype=np.float64)
        self.grid_center = np.ascontiguousarray(self.grid_center, dtype=np.float64)
        c = ctypes
        self._lib._get_observation(
            _dbl(self.p), _dbl(self.dp), _dbl(self.heading), _dbl(self.obs), _dbl(self.boundary_pos),
            _dbl(self.grid_center), _i32(self.neighbor_index), _i32(self.in_flags), _i32(self.sensed_index),
            _i32(self.occupied_index), c.c_double(self.d_sen), c.c_double(self.r_avoid), c.c_double(self.l_cell),
            c.c_double(self.Vel_max), c.c_int(self.topo_nei_max), c.c_int(self.num_obs_grid_max),
            c.c_int(self.num_occupied_grid_max), c.c_int(self.n_a), c.c_int(self.n_g), c.c_int(self.obs_dim_agent),
            c.c_int(self.dim), _bool(cond))
        _lib.check_compat("_get_observation")          # the C symbol is void and never ends the process; this class raises
        return self.obs

    def _get_reward(self, a):
        """ENV:346-375 -> _get_reward."""
        reward_a = np.zeros((1, self.n_a))
        coefficients = np.array([0.05], dtype=np.float64)
        cond = np.array([self.is_periodic, self.is_Cartesian, self.penalize_entering, self.penalize_interaction,
                         self.penalize_exploration], dtype=bool)
        act = np.ascontiguousarray(a, dtype=np.float64)
        c = ctypes
        self._lib._get_reward(
            _dbl(self.p), _dbl(self.dp), _dbl(self.heading), _dbl(act), _dbl(reward_a), _dbl(self.boundary_pos),
            _dbl(self.grid_center), _i32(self.neighbor_index), _i32(self.in_flags), _i32(self.sensed_index),
            _i32(self.occupied_index), c.c_double(self.d_sen), c.c_double(self.r_avoid), c.c_double(self.l_cell),
            c.c_int(self.topo_nei_max), c.c_int(self.num_obs_grid_max), c.c_int(self.num_occupied_grid_max),
            c.c_int(self.n_a), c.c_int(self.n_g), c.c_int(self.dim), _bool(cond),
            _bool(np.ascontiguousarray(self.is_collide_b2b)), _bool(self.is_collide_b2w), _dbl(coefficients))
        _lib.check_compat("_get_reward")
        return reward_a

    def _get_dist_b2b(self):
        """ENV:437-452.  The periodic wrap reaches agent 0's row only (ENV:773-776 indexes
        rows 0 and 1 of the stacked (2n, n) matrix); that is kept."""
        rel = self.p[:, None, :] - self.p[:, :, None]                 # rel[c, i, j] = p[c, j] - p[c, i]
        if self.is_periodic:
            rel[:, 0, :] = self._make_periodic(rel[:, 0, :], is_rel=True)
        d_center = np.sqrt(rel[0] ** 2 + rel[1] ** 2)
        gap = d_center - self.sizes
        self.d_b2b_center = d_center
        self.is_collide_b2b = gap < 0
        self.d_b2b_edge = np.abs(gap)
        return self.d_b2b_center, self.d_b2b_edge, self.is_collide_b2b

    def _get_dist_b2w(self):
        """ENV:454-461 -> _get_dist_b2w."""
        self._lib._get_dist_b2w(_dbl(self.p), _dbl(self.size), _dbl(self.d_b2w), _bool(self.is_collide_b2w),
                                ctypes.c_int(self.dim), ctypes.c_int(self.n_a), _dbl(self.boundary_pos))
        _lib.check_compat("_get_dist_b2w")

    def _get_done(self):
        return np.zeros((1, self.n_a)).astype(bool)                   # ENV:475-477

    def _get_info(self):
        return np.array([None, None, None]).reshape(3, 1)             # ENV:479-480

    # ------------------------------------------------------------------ step
    def step(self, a):
        """ENV:482-661."""
        self.simulation_time += self.dt
        for _ in range(self.n_frames):
            d_center, d_edge, collide = self._get_dist_b2b()
            sf_b2b = np.zeros((2, self.n_a))
            self._lib._sf_b2b_all(_dbl(self.p), _dbl(sf_b2b), _dbl(np.ascontiguousarray(d_edge)),
                                  _bool(np.ascontiguousarray(collide)), _dbl(self.boundary_pos),
                                  _dbl(np.ascontiguousarray(d_center)), ctypes.c_int(self.n_a), ctypes.c_int(self.dim),
                                  ctypes.c_double(self.k_ball), ctypes.c_bool(self.is_periodic))
            _lib.check_compat("_sf_b2b_all")
            if self.is_boundary:
                self._get_dist_b2w()
                hit = self.is_collide_b2w * self.d_b2w                                   # ENV:512
                sf_b2w = np.array([[1, 0, -1, 0], [0, -1, 0, 1]]).dot(hit) * self.k_wall
                vel4 = self.is_collide_b2w * np.concatenate((self.dp, self.dp), axis=0)   # ENV:513
                df_b2w = np.array([[-1, 0, -1, 0], [0, -1, 0, -1]]).dot(vel4) * self.c_wall
            if self.agent_strategy == "input":
                pass
            elif self.agent_strategy == "random":
                a = np.random.uniform(-1, 1, (self.act_dim_agent, self.n_a))             # ENV:519
            elif self.agent_strategy == "llm":                                           # ENV:520-524
                a = np.zeros((self.dim, self.n_a))
                for i in range(self.n_a):
                    a[:, i] = self.robot_prior_policy(i, self._get_trgt_grid_state(i)[1])
            elif self.agent_strategy == "rule":                                          # ENV:525-596
                a = np.zeros((self.dim, self.n_a))
                for i in range(self.n_a):
                    a[:, i] = self._rule_action(i)
            else:
                print("Wrong in Step function")                                          # ENV:597-598
            a_prior = None
            if self.training_method == "llm_rl":
                a_prior = np.zeros((self.dim, self.n_a))
                self._lib.calculateActionPrior(
                    _dbl(self.p), _dbl(self.dp), _dbl(a_prior), _dbl(self.grid_center), _i32(self.neighbor_index),
                    ctypes.c_double(self.d_sen), ctypes.c_double(self.r_avoid), ctypes.c_double(self.l_cell),
                    ctypes.c_int(self.topo_nei_max), ctypes.c_int(self.n_a), ctypes.c_int(self.n_g), ctypes.c_int(self.dim))
                _lib.check_compat("calculateActionPrior")
            if self.dynamics_mode == "Cartesian":
                u = a
            else:
                print("Wrong in updating dynamics")
            F = self.sensitivity * u + sf_b2b                                            # ENV:632-635
            if self.is_boundary:
                F = F + sf_b2w + df_b2w
            self.ddp = F / self.m
            self.dp += self.ddp * self.dt
            self.dp = np.clip(self.dp, -self.Vel_max, self.Vel_max)
            self.p += self.dp * self.dt
            if self.is_periodic:
                self.p = self._make_periodic(self.p, is_rel=False)
            obs = self._get_obs()
            rew = self._get_reward(a)
            done = self._get_done()
            info = self._get_info()
        if self.is_collected:
            return obs, rew, done, info, u
        return obs, rew, done, info, (a_prior if self.training_method == "llm_rl" else None)

    def render(self, mode="human"):
        raise NotImplementedError("rendering (ENV:663-752, matplotlib) is not part of this build")

    # ------------------------------------------------------------------ in-environment policies
    def _get_trgt_grid_state(self, self_id):
        """ENV:833-849: in-shape flag, target position/velocity and the cells inside the sensing range."""
        norms = np.linalg.norm(self.grid_center - self.p[:, [self_id]], axis=0)
        nearest = np.argmin(norms)
        sensed = np.where(norms < self.d_sen)[0]
        if norms[nearest] < np.sqrt(2) * self.l_cell / 2:
            return 1, self.p[:, self_id], self.dp[:, self_id], sensed
        return 0, self.grid_center[:, nearest], np.array([0, 0]), sensed

    def _rho_cos_dec(self, z, delta, r):
        """ENV:867-870."""
        z = np.asarray(z)
        return np.where(z < delta * r, 1.0, np.where(z < r, 0.5 * (1.0 + np.cos(np.pi * (z / r - delta) / (1.0 - delta))), 0.0))

    def robot_prior_policy(self, robot_id, target_position):
        """ENV:897-944: attraction to the target, repulsion from listed neighbours closer than r_avoid,
        velocity synchronisation with the listed neighbours, clipped to [-1, 1]."""
        pos, vel = self.p[:, robot_id], self.dp[:, robot_id]
        force = np.array([0.0, 0.0])
        to_target = target_position - pos
        dist = np.linalg.norm(to_target)
        force += 2.0 * to_target / dist if dist > 0 else np.array([0.0, 0.0])
        nbrs = self.neighbor_index[robot_id]
        nbrs = nbrs[nbrs != -1]
        vels = []
        for j in nbrs:
            away = pos - self.p[:, j]
            d = np.linalg.norm(away)
            vels.append(self.dp[:, j])
            if 0 < d < self.r_avoid:
                force += 1.0 * (self.r_avoid / d - 1) * (away / d)
        if vels:
            force += 2.0 * (np.mean(vels, axis=0) - vel)
        return np.clip(force, -1, 1)

    def _rule_action(self, i):
        """ENV:528-594 for agent i: entering + exploration + interaction velocities, clipped."""
        k_1, k_2, k_3 = 1, 15, 17
        in_flag, t_pos, t_vel, sensed = self._get_trgt_grid_state(i)
        to_t = t_pos - self.p[:, i]
        v_ent = np.zeros(self.dim) if in_flag == 1 else \
            k_1 * (to_t / (np.linalg.norm(to_t) + 1e-8)) + (t_vel - self.dp[:, i])
        rel = self.p - self.p[:, [i]]
        rel_norm = np.linalg.norm(rel, axis=0)
        if len(sensed) > 0 and in_flag == 1:                           # drop the cells some nearby agent occupies
            cells = self.grid_center[:, sensed]
            for j in np.where(rel_norm < (self.d_sen + self.r_avoid / 2))[0]:
                keep = np.where(np.linalg.norm(cells - self.p[:, [j]], axis=0) > self.r_avoid / 2)[0]
                cells = cells[:, keep]
                sensed = sensed[keep]
        n = len(sensed)
        if n > self.num_obs_grid_max:                                  # ENV:560-565
            pick = np.round(np.arange(0, self.num_obs_grid_max) * ((n - 1) / (self.num_obs_grid_max - 1))).astype(int)
            sensed = np.array(sensed)[pick]
        v_exp = np.zeros(self.dim)
        if n > 0:
            g = self.grid_center[:, sensed] - self.p[:, [i]]
            psi = self._rho_cos_dec(np.linalg.norm(g, axis=0), 0, self.d_sen)
            den = np.sum(psi)
            if den == 0:
                den = 1e-8
            v_exp += k_2 * np.sum(psi * g, axis=1) / den
        v_int = np.zeros(self.dim)
        near = np.where(rel_norm < self.d_sen)[0]
        near = near[near != i]
        vel_rel = self.dp - self.dp[:, [i]]
        for j in near:
            if rel_norm[j] < self.r_avoid:
                v_int += -k_3 * (self.r_avoid / rel_norm[j] - 1) * rel[:, j]
            v_int += 5 * vel_rel[:, j] / len(near)
        return np.clip(v_ent + v_exp + v_int, -1, 1)

    # ------------------------------------------------------------------ helpers
    def _make_periodic(self, x, is_rel):
        """ENV:771-782."""
        W, H = self.boundary_width_half, self.boundary_height_half
        if is_rel:
            x[0, x[0, :] < -W] += 2 * W
            x[0, x[0, :] > W] -= 2 * W
            x[1, x[1, :] < -H] += 2 * H
            x[1, x[1, :] > H] -= 2 * H
        else:
            x[0, x[0, :] < self.boundary_pos[0]] += 2 * W
            x[0, x[0, :] > self.boundary_pos[2]] -= 2 * W
            x[1, x[1, :] < self.boundary_pos[3]] += 2 * H
            x[1, x[1, :] > self.boundary_pos[1]] -= 2 * H
        return x
