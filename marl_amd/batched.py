"""BatchedAssemblyEnv — E independent assembly environments stepped by one HIP
kernel launch, behind the reset()/step() interface of the reference's
AssemblySwarmEnv (cus_gym/gym/envs/customized_envs/assembly.py:151, 482).

PyTorch is used for device memory and streams only; all environment arithmetic
runs in libAssemblyEnv.so (marl_amd/csrc) through the C ABI.

Tensor shapes keep the reference's per-environment layouts with a leading batch
axis: obs (E, obs_dim, n_a), actions (E, 2, n_a), reward (E, 1, n_a),
done (E, 1, n_a) all False, a_prior (E, 2, n_a).
"""
from __future__ import annotations

import ctypes as C
import os
from types import SimpleNamespace
from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib
from .shapes import load_shapes


def r_avoid_for(n_a: int, n_gs: Sequence[int], l_cells: Sequence[float]) -> float:
    """assembly.py:121-122."""
    return round(float(np.sqrt(4 * np.min(n_gs) / (n_a * np.pi)) * np.min(l_cells)), 2)


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


# the attributes step() hands to the kernel: re-pointing one of them (DeviceReplay and StreamShards do that with the
# outputs) drops the cached argument list
_STEP_INPUTS = frozenset(("obs", "reward", "a_prior", "p", "dp", "grid_center", "n_g", "l_cell", "neighbor_index", "in_flags",
                          "nearest_cell", "sensed_index", "occupied_index", "stats", "shape_id", "xform", "training_method",
                          "_lattice", "params", "E"))


class BatchedAssemblyEnv:
    def __setattr__(self, name, value):
        if name in _STEP_INPUTS:
            object.__setattr__(self, "_step_args", None)
        object.__setattr__(self, name, value)

    def __init__(self, num_envs: int, n_a: int = 30, *, device="cuda", seed: int = 226, env_id_offset: int = 0,
                 is_boundary: bool = True, is_con_self_state: bool = True, training_method: str = "llm_rl",
                 shapes=None, r_avoid: Optional[float] = None, d_sen: float = 0.4, n_g_max: Optional[int] = None,
                 keep_indices=False, keep_stats: bool = True, num_obs_grid_max: int = 80,
                 num_occupied_grid_max: int = 200, use_lattice: bool = True):
        self.lib = _lib.load()
        self.E = int(num_envs)
        self.n_a = int(n_a)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("BatchedAssemblyEnv runs on a HIP device only (device='cuda'); there is no CPU path")
        self.seed = int(seed)
        self.env_id_offset = int(env_id_offset)
        self.episode = -1
        self.step_count = 0
        l_cells, grids = shapes if shapes is not None else load_shapes()
        self.l_cells = list(l_cells)
        self.grid_center_origins = [np.asarray(g, dtype=np.float64) for g in grids]
        self.n_gs = [g.shape[0] for g in self.grid_center_origins]
        self.num_train_shape = len(self.l_cells)
        self.r_avoid = r_avoid if r_avoid is not None else r_avoid_for(self.n_a, self.n_gs, self.l_cells)
        self.n_g_max = int(n_g_max) if n_g_max is not None else (max(self.n_gs) + 31) // 32 * 32
        self.training_method = training_method
        self.params = _lib.default_params(self.n_a, self.n_g_max, self.r_avoid, is_boundary=int(is_boundary),
                                          is_con_self_state=int(is_con_self_state), d_sen=d_sen,
                                          compute_prior=int(training_method == "llm_rl"),
                                          num_obs_grid_max=int(num_obs_grid_max),            # assembly.py:126
                                          num_occupied_grid_max=int(num_occupied_grid_max))  # assembly.py:128
        self.obs_dim_agent = self.lib.asm_obs_dim(C.byref(self.params))
        # spaces as the learner reads them (assembly.py:806-812; maddpg.py:264-266 reads .shape)
        self.observation_space = SimpleNamespace(shape=(self.obs_dim_agent, self.n_a), dtype=np.float32)
        self.action_space = SimpleNamespace(shape=(2, self.n_a), dtype=np.float32)
        self.num_agents = self.n_a
        self.agent_types = ["agent"]
        self.alpha = 1

        E, n, dev = self.E, self.n_a, self.device
        f32 = dict(dtype=torch.float32, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        self.p = torch.zeros(E, 2, n, **f32)
        self.dp = torch.zeros(E, 2, n, **f32)
        self.grid_center = torch.zeros(E, 2, self.n_g_max, **f32)
        self.n_g = torch.zeros(E, **i32)
        self.l_cell = torch.zeros(E, dtype=torch.float64, device=dev)
        self.shape_id = torch.zeros(E, **i32)
        self.neighbor_index = torch.full((E, n, 6), -1, **i32)
        self.in_flags = torch.zeros(E, n, **i32)
        self.nearest_cell = torch.zeros(E, n, **i32)
        self.obs = torch.zeros(E, self.obs_dim_agent, n, **f32)
        self.reward = torch.zeros(E, 1, n, **f32)
        self.a_prior = torch.zeros(E, 2, n, **f32)
        self.done = torch.zeros(E, 1, n, dtype=torch.bool, device=dev)          # assembly.py:475-477
        # keep_indices: True = both index lists of assembly.py:225-226, "sensed" = the sensed list only (what
        # rule_actions needs; the occupied list costs far more to write), False = neither
        if keep_indices not in (True, False, "sensed"):
            raise ValueError('keep_indices is True, False or "sensed"')
        self.sensed_index = torch.full((E, n, int(num_obs_grid_max)), -1, **i32) if keep_indices else None
        self.occupied_index = torch.full((E, n, int(num_occupied_grid_max)), -1, **i32) if keep_indices is True else None
        self.stats = torch.zeros(E, 8, dtype=torch.float64, device=dev) if keep_stats else None
        # canonical shapes on the device (results.pkl 'grid_coords' layout, assembly.py:115)
        off = np.zeros(self.num_train_shape, np.int32)
        off[1:] = np.cumsum(self.n_gs)[:-1]
        self._shape_grids = torch.from_numpy(np.concatenate(self.grid_center_origins, 0)).to(dev)
        self._shape_off = torch.from_numpy(off).to(dev)
        self._shape_ng = torch.tensor(self.n_gs, **i32)
        self._shape_lcell = torch.tensor(self.l_cells, dtype=torch.float64, device=dev)
        self._recompute_target = False
        self._step_args, self._step_prior, self._lat_ref = None, None, None
        # The shapes' lattice description (include/marl_amd_env.h, asm_lattice): lets the step kernel enumerate lattice
        # rows instead of testing every cell.  It describes grids produced by reset() / set_shape(); it is switched
        # off until the first of those, and by invalidate_lattice() (call it after writing grid_center directly).
        self.xform = torch.zeros(E, 4, dtype=torch.float64, device=dev)
        self._lattice, self._lattice_shapes, self.lattice_valid = None, None, False
        if use_lattice and os.environ.get("MARL_AMD_NO_LATTICE") != "1":     # the variable: A/B measurements of one build
            self._build_lattice()
        self._dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()

    # ------------------------------------------------------------------
    def _build_lattice(self) -> None:
        arr = (_lib.AsmLatticeShape * self.num_train_shape)()
        for k, g in enumerate(self.grid_center_origins):
            xy = np.ascontiguousarray(g, dtype=np.float64)
            if self.lib.asm_lattice_build(xy.ctypes.data, xy.shape[0], float(self.l_cells[k]), C.byref(arr[k])) != 0:
                return                                          # not a lattice (custom shapes): every cell is tested
        raw = np.frombuffer(bytes(arr), dtype=np.uint8).copy()
        self._lattice_shapes = torch.from_numpy(raw).to(self.device)
        self._lattice = _lib.AsmLattice(self._lattice_shapes.data_ptr(), self.num_train_shape, 0, self.shape_id.data_ptr(),
                                        self.xform.data_ptr(), float(min(self.l_cells)))

    def _lat(self):
        return C.byref(self._lattice) if (self._lattice is not None and self.lattice_valid) else None

    def step_schedule(self) -> int:
        """Which kernel step() launches now (asm_debug_step_schedule): waves per workgroup of the multi-wave kernel or
        0 for the generic one, + 16 with its lattice path, + 32 with the generic kernel's lattice path."""
        return self.lib.asm_debug_step_schedule(C.byref(self.params), self.E, self._lat())

    def invalidate_lattice(self) -> None:
        """Call after writing grid_center / n_g / shape_id / xform directly: the kernels then test every cell against
        every agent (the grids no longer follow the lattice description) until the next reset() or set_shape()."""
        self.lattice_valid = False

    def check_lattice(self, tol: float = 1e-6) -> int:
        """Cells of grid_center that are not where the lattice description says (0 when it is valid); synchronises."""
        if self._lattice is None:
            raise RuntimeError("no lattice description (use_lattice=False or the shapes are not lattices)")
        n_bad = torch.zeros(1, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            rc = self.lib.asm_batch_check_lattice(C.byref(self.params), self.E, _ptr(self.grid_center), _ptr(self.n_g),
                                                  C.byref(self._lattice), float(tol), _ptr(n_bad), self._stream())
        _lib.check(rc, "asm_batch_check_lattice")
        return int(n_bad.item())

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def reset(self, episode: Optional[int] = None) -> torch.Tensor:
        """assembly.py:151-218 for every environment; returns obs (E, obs_dim, n_a)."""
        self.episode = self.episode + 1 if episode is None else int(episode)
        self.step_count = 0
        with torch.cuda.device(self.device):
            rc = self.lib.asm_batch_reset_f32(
                C.byref(self.params), self.seed, self.episode, self.env_id_offset, self.E, self.num_train_shape,
                _ptr(self._shape_grids), _ptr(self._shape_off), _ptr(self._shape_ng), _ptr(self._shape_lcell),
                _ptr(self.shape_id), _ptr(self.p), _ptr(self.dp), _ptr(self.grid_center), _ptr(self.n_g),
                _ptr(self.l_cell), _ptr(self.neighbor_index), _ptr(self.in_flags), _ptr(self.nearest_cell),
                _ptr(self.obs), _ptr(self.sensed_index), _ptr(self.occupied_index), _ptr(self.xform),
                None if self._lattice is None else C.byref(self._lattice), self._stream())
        _lib.check(rc, "asm_batch_reset_f32")
        self.lattice_valid = self._lattice is not None
        if self.stats is not None:
            self.stats.zero_()
        self._recompute_target = False
        return self.obs

    def observe(self, want_reward: bool = False) -> torch.Tensor:
        """Recompute obs / neighbour lists / flags from the current tensors (after the
        caller wrote p, dp, grid_center, n_g or l_cell directly)."""
        with torch.cuda.device(self.device):
            rc = self.lib.asm_batch_observe_f32(
                C.byref(self.params), self.E, _ptr(self.p), _ptr(self.dp), _ptr(self.grid_center), _ptr(self.n_g),
                _ptr(self.l_cell), _ptr(self.neighbor_index), _ptr(self.in_flags), _ptr(self.nearest_cell),
                _ptr(self.obs), _ptr(self.reward) if want_reward else None, _ptr(self.sensed_index),
                _ptr(self.occupied_index), self._lat(), self._stream())
        _lib.check(rc, "asm_batch_observe_f32")
        self._recompute_target = False
        return self.obs

    def invalidate_target_cache(self) -> None:
        """Call after changing p or the grid between steps without observe(): the next
        step's prior then redoes its own nearest-cell search (AssemblyEnv.cpp:1107)."""
        self._recompute_target = True

    def step(self, a: torch.Tensor, stream_ptr: Optional[int] = None):
        """assembly.py:482-661 ('input' strategy).  a: (E, 2, n_a) float32 on the device.
        Returns (obs, reward, done, info, a_prior) like the reference's 5-tuple.
        stream_ptr: the HIP stream to launch on (default: torch's current stream on this device)."""
        if a.dtype != torch.float32 or not a.is_contiguous() or a.device != self.p.device or \
                tuple(a.shape) != (self.E, 2, self.n_a):
            raise ValueError(f"actions must be a contiguous float32 tensor of shape {(self.E, 2, self.n_a)} on {self.device}")
        # The argument list only changes when one of the attributes it is built from is assigned (__setattr__ drops it;
        # DeviceReplay and StreamShards re-point the outputs that way), so `env.x = other_tensor` is always honoured;
        # building 17 ctypes pointers per call, or even reading 16 data pointers to see whether anything moved, costs more
        # host time than the rest of the call, and three or four shards per step make the host the limit (10.8 us per
        # call before, against a 31 us step).  Pointers go in as plain integers (the prototypes are void *).
        args = self._step_args
        if args is None:
            prior = self.a_prior if self.training_method == "llm_rl" else None
            dp_ = lambda t: None if t is None else t.data_ptr()
            args = [C.byref(self.params), self.E, dp_(self.p), dp_(self.dp), None, dp_(self.grid_center),
                    dp_(self.n_g), dp_(self.l_cell), dp_(self.neighbor_index), dp_(self.in_flags),
                    dp_(self.nearest_cell), dp_(self.obs), dp_(self.reward), dp_(prior),
                    dp_(self.sensed_index), dp_(self.occupied_index), dp_(self.stats), 0, None, None]
            if self._lattice is not None:
                self._lattice.shape_id, self._lattice.xform = self.shape_id.data_ptr(), self.xform.data_ptr()
            d = self.__dict__
            d["_step_args"], d["_step_prior"] = args, prior
            d["_lat_ref"] = None if self._lattice is None else C.byref(self._lattice)
        d = self.__dict__
        args[4] = a.data_ptr(); args[17] = 1 if d["_recompute_target"] else 0
        args[18] = d["_lat_ref"] if d["lattice_valid"] else None
        args[19] = torch.cuda.current_stream(self.device).cuda_stream if stream_ptr is None else stream_ptr
        if torch.cuda.current_device() == self._dev_index:
            rc = self.lib.asm_batch_step_f32(*args)
        else:
            with torch.cuda.device(self.device):
                rc = self.lib.asm_batch_step_f32(*args)
        if rc:
            _lib.check(rc, "asm_batch_step_f32")
        d["_recompute_target"] = False
        d["step_count"] += 1
        return self.obs, self.reward, self.done, None, d["_step_prior"]

    def random_actions(self, out: Optional[torch.Tensor] = None, step: Optional[int] = None) -> torch.Tensor:
        """U(-1,1) actions from the counter-based generator (assembly.py:519)."""
        if out is None:
            out = torch.empty(self.E, 2, self.n_a, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            rc = self.lib.asm_batch_random_actions_f32(self.seed, self.step_count if step is None else int(step),
                                                       self.env_id_offset, self.E, self.n_a, _ptr(out), self._stream())
        _lib.check(rc, "asm_batch_random_actions_f32")
        return out

    def rule_actions(self, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """(E, 2, n_a) actions of the reference's 'rule' expert (assembly.py:525-596) for the current state, i.e.
        the state of the last reset()/step()/observe().  Needs keep_indices=True (the sensed lists)."""
        if self.sensed_index is None:
            raise RuntimeError('rule_actions needs the sensed-index lists: construct with keep_indices=True or "sensed"')
        if out is None:
            out = torch.empty(self.E, 2, self.n_a, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            rc = self.lib.asm_batch_rule_actions_f32(
                C.byref(self.params), self.E, _ptr(self.p), _ptr(self.dp), _ptr(self.grid_center), _ptr(self.n_g),
                _ptr(self.in_flags), _ptr(self.nearest_cell), _ptr(self.sensed_index), _ptr(out), self._stream())
        _lib.check(rc, "asm_batch_rule_actions_f32")
        return out

    METRIC_FIELDS = ("coverage_rate", "distribution_uniformity", "voronoi_uniformity", "num_collisions",
                     "avg_distance_to_target")

    def formation_metrics(self, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """(E, 5) float64: the wrapper's get_formation_metrics (assembly_wrapper.py:167-187) of the current
        positions of every environment, columns as METRIC_FIELDS.  num_collisions counts the contact pairs of
        the current positions (the reference's count reads the matrix its last step stored, i.e. the positions
        that step started from: call this before step() to get that number)."""
        if out is None:
            out = torch.empty(self.E, 5, dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            rc = self.lib.asm_batch_metrics_f64(C.byref(self.params), self.E, _ptr(self.p), _ptr(self.grid_center),
                                                _ptr(self.n_g), _ptr(out), self._stream())
        _lib.check(rc, "asm_batch_metrics_f64")
        return out

    def set_shape(self, shape_index: int, env_ids: Optional[torch.Tensor] = None) -> None:
        """Switch environments (all, or the given ids) to canonical shape `shape_index`, unrotated and centred,
        keeping the agents where they are: eval_assembly.py:17-57 (process_shape).  The next step's prior then
        searches its target cell on the new grid (AssemblyEnv.cpp:1107), as the reference's does every step."""
        k = int(shape_index)
        if not 0 <= k < self.num_train_shape:
            raise IndexError(f"shape {k}: {self.num_train_shape} shapes are loaded")
        n_g = self.n_gs[k]
        if n_g > self.n_g_max:
            raise ValueError(f"shape {k} has {n_g} cells, the grid arrays hold {self.n_g_max}")
        o = int(self._shape_off[k])
        row = torch.zeros(2, self.n_g_max, dtype=torch.float32, device=self.device)   # cells beyond n_g are never read
        row[:, :n_g] = self._shape_grids[o:o + n_g].T.to(torch.float32)
        sel = slice(None) if env_ids is None else env_ids.to(self.device, torch.long)
        self.grid_center[sel] = row
        self.n_g[sel] = n_g
        self.l_cell[sel] = self.l_cells[k]
        self.shape_id[sel] = k
        self.xform[sel] = torch.tensor([1.0, 0.0, 0.0, 0.0], dtype=torch.float64, device=self.device)   # unrotated, centred
        if env_ids is None:                                     # every grid now follows the lattice description again
            self.lattice_valid = self._lattice is not None
        self._recompute_target = True

    def episode_stats(self) -> torch.Tensor:
        """[sum r, sum r^2, sum coverage, sum uniformity, sum contacts, sum avg-distance, steps, sum inside]
        summed over this rank's environments (float64, 8 values) — the vector that is
        all-reduced across GPUs once per episode."""
        if self.stats is None:
            raise RuntimeError("constructed with keep_stats=False")
        return self.stats.sum(dim=0)
