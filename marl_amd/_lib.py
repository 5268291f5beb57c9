"""ctypes binding of marl_amd/csrc/build/libAssemblyEnv.so (include/marl_amd_env.h).

The library is the HIP implementation; there is no Python or CPU fallback.  If it
has not been built, loading raises with the build command.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MARL_AMD_LIB: load another build of the same library (e.g. the diagnostic build of `make diag`)
SO_PATH = os.environ.get("MARL_AMD_LIB") or os.path.join(_HERE, "csrc", "build", "libAssemblyEnv.so")

ABI_VERSION = 2


class AsmParams(C.Structure):
    """struct asm_params of include/marl_amd_env.h."""
    _fields_ = [("n_a", C.c_int32), ("n_g_max", C.c_int32), ("topo_nei_max", C.c_int32),
                ("num_obs_grid_max", C.c_int32), ("num_occupied_grid_max", C.c_int32),
                ("is_boundary", C.c_int32), ("is_con_self_state", C.c_int32), ("compute_prior", C.c_int32),
                ("size_a", C.c_double), ("d_sen", C.c_double), ("r_avoid", C.c_double), ("vel_max", C.c_double),
                ("k_ball", C.c_double), ("k_wall", C.c_double), ("c_wall", C.c_double), ("dt", C.c_double),
                ("sensitivity", C.c_double), ("mass", C.c_double), ("half_w", C.c_double), ("half_h", C.c_double)]


LAT_ROWS, LAT_CHUNKS, LAT_NB = 48, 3, 256


class AsmLatticeShape(C.Structure):
    """struct asm_lattice_shape of include/marl_amd_env.h."""
    _fields_ = [("row_tab", C.c_uint32 * 2 * LAT_CHUNKS * LAT_ROWS), ("boundary", C.c_uint16 * LAT_NB),
                ("n_boundary", C.c_int32), ("cols", C.c_int32), ("rows", C.c_int32), ("n_g", C.c_int32),
                ("x0", C.c_double), ("y0", C.c_double), ("l_cell", C.c_double), ("reserved", C.c_double)]


class AsmLattice(C.Structure):
    """struct asm_lattice of include/marl_amd_env.h (device pointers as integers)."""
    _fields_ = [("shapes", C.c_void_p), ("n_shapes", C.c_int32), ("reserved", C.c_int32), ("shape_id", C.c_void_p),
                ("xform", C.c_void_p), ("l_cell_min", C.c_double)]


# Every symbol include/marl_amd_env.h declares.
EXPORTS = ["_get_observation", "_get_reward", "_sf_b2b_all", "_get_dist_b2w", "calculateActionPrior",
           "asm_default_params", "asm_obs_dim", "asm_abi_version", "asm_last_error", "asm_compat_status", "asm_device_count",
           "asm_batch_observe_f32", "asm_batch_step_f32", "asm_batch_reset_f32",
           "asm_batch_random_actions_f32", "asm_batch_metrics_f64", "asm_batch_rule_actions_f32", "asm_replay_gather_f32", "asm_replay_draw_indices",
           "asm_debug_sqrt_f64", "asm_debug_cos_f32_max_err", "asm_get_observation_noalloc", "asm_lattice_build",
           "asm_batch_check_lattice", "asm_set_step_schedule", "asm_debug_step_schedule",
           "asm_policy_mlp_forward_bf16", "asm_policy_mlp_act_bf16", "asm_obs_rows_f32"]

_lib = None


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(SO_PATH):
        raise RuntimeError(
            f"{SO_PATH} is missing: the HIP library has not been built. Run "
            "`make -C marl_amd/csrc` (or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "marl_amd has no CPU fallback for the environment step.")
    try:  # share torch's HIP runtime when torch is present (same SONAME, load it first)
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(SO_PATH, C.RTLD_GLOBAL)
    lib.asm_last_error.restype = C.c_char_p
    lib.asm_abi_version.restype = C.c_int
    lib.asm_compat_status.restype = C.c_int
    lib.asm_device_count.restype = C.c_int
    lib.asm_obs_dim.restype = C.c_int
    lib.asm_obs_dim.argtypes = [C.POINTER(AsmParams)]
    lib.asm_default_params.restype = None
    lib.asm_default_params.argtypes = [C.POINTER(AsmParams), C.c_int, C.c_int]
    vp, i32, i64, u32, u64 = C.c_void_p, C.c_int, C.c_int64, C.c_uint32, C.c_uint64
    PP = C.POINTER(AsmParams)
    lib.asm_batch_observe_f32.restype = C.c_int
    LP = C.POINTER(AsmLattice)
    lib.asm_batch_observe_f32.argtypes = [PP, i32] + [vp] * 12 + [LP, vp]
    lib.asm_batch_step_f32.restype = C.c_int
    lib.asm_batch_step_f32.argtypes = [PP, i32] + [vp] * 15 + [i32, LP, vp]
    lib.asm_batch_reset_f32.restype = C.c_int
    lib.asm_batch_reset_f32.argtypes = [PP, u64, u32, i64, i32, i32] + [vp] * 16 + [vp, LP, vp]
    lib.asm_lattice_build.restype = C.c_int
    lib.asm_lattice_build.argtypes = [vp, i32, C.c_double, C.POINTER(AsmLatticeShape)]
    lib.asm_batch_check_lattice.restype = C.c_int
    lib.asm_batch_check_lattice.argtypes = [PP, i32, vp, vp, LP, C.c_double, vp, vp]
    lib.asm_batch_random_actions_f32.restype = C.c_int
    lib.asm_batch_random_actions_f32.argtypes = [u64, u32, i64, i32, i32, vp, vp]
    lib.asm_batch_rule_actions_f32.restype = C.c_int
    lib.asm_batch_rule_actions_f32.argtypes = [PP, i32] + [vp] * 8 + [vp]
    lib.asm_batch_metrics_f64.restype = C.c_int
    lib.asm_batch_metrics_f64.argtypes = [PP, i32, vp, vp, vp, vp, vp]
    lib.asm_replay_gather_f32.restype = C.c_int
    lib.asm_replay_gather_f32.argtypes = [i32, i32, i32, i32] + [vp] * 5 + [i32] + [vp] * 5 + [vp]
    lib.asm_replay_draw_indices.restype = C.c_int
    lib.asm_replay_draw_indices.argtypes = [u64, u32, vp, i64, i64, i32, vp, vp]
    lib.asm_debug_sqrt_f64.restype = C.c_int
    lib.asm_debug_sqrt_f64.argtypes = [vp, vp, i32, vp]
    lib.asm_debug_cos_f32_max_err.restype = C.c_int
    lib.asm_debug_cos_f32_max_err.argtypes = [u32, u32, vp, vp]
    lib.asm_get_observation_noalloc.restype = C.c_int
    lib.asm_debug_step_schedule.restype = C.c_int
    lib.asm_debug_step_schedule.argtypes = [PP, i32, LP]
    lib.asm_policy_mlp_forward_bf16.restype = C.c_int
    lib.asm_policy_mlp_forward_bf16.argtypes = [i32] * 5 + [vp] * 8
    lib.asm_policy_mlp_act_bf16.restype = C.c_int
    lib.asm_policy_mlp_act_bf16.argtypes = [i32] * 5 + [vp] * 7 + [C.c_float, C.c_float, u64, u32, i64, vp]
    lib.asm_obs_rows_f32.restype = C.c_int
    lib.asm_obs_rows_f32.argtypes = [i32, i32, i32, vp, vp, vp]
    lib.asm_set_step_schedule.restype = None
    lib.asm_set_step_schedule.argtypes = [C.c_int]
    if lib.asm_abi_version() != ABI_VERSION:
        raise RuntimeError(f"libAssemblyEnv.so ABI {lib.asm_abi_version()} != expected {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"{what} failed ({rc}): {load().asm_last_error().decode()}")


def check_compat(what: str) -> None:
    """After one of the five reference-named void calls: raise if it failed (its outputs are then NaN / -1)."""
    check(load().asm_compat_status(), what)


def default_params(n_a: int, n_g_max: int, r_avoid: float, **over) -> AsmParams:
    p = AsmParams()
    load().asm_default_params(C.byref(p), n_a, n_g_max)
    p.r_avoid = r_avoid
    for k, v in over.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p
