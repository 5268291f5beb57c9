"""TEST INFRASTRUCTURE — loads the *reference* Python environment for oracle use.

Only usable where /root/reference exists (the build container).  Nothing here is
imported by the product package; tests/golden/make_golden.py and the `not gpu`
oracle-validation tests are the only callers.

What it does:
  * builds nothing itself: expects oracle/_ref/libAssemblyEnv.so (see oracle/Makefile),
    compiled from the reference's own AssemblyEnv.cpp where it lies;
  * puts the reference's `cus_gym` on sys.path (no other gym is installed);
  * redirects the ctypes load done by envs_cplus/c_lib.py:11-45 (which looks for
    build/libAssemblyEnv.so inside the read-only tree) to oracle/_ref/;
  * writes a results.pkl with the keys assembly.py:110-117 reads, from the grids
    of marl_amd/data/shapes.npz (our own PIL+numpy restatement of the cfg image
    pipeline), into a temp dir.
"""
from __future__ import annotations

import ctypes
import os
import pickle
import sys
import tempfile
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF_ROOT = "/root/reference"
REF_SO = os.path.join(HERE, "_ref", "libAssemblyEnv.so")


def available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "cus_gym")) and os.path.isfile(REF_SO)


_gym = None


def _import_reference_gym():
    global _gym
    if _gym is not None:
        return _gym
    if not available():
        raise RuntimeError("reference tree or oracle/_ref/libAssemblyEnv.so missing")
    sys.path.insert(0, os.path.join(REF_ROOT, "cus_gym"))
    real_cdll = ctypes.CDLL

    class _CDLL(real_cdll):
        def __init__(self, name, *a, **k):
            if name and "libAssemblyEnv" in str(name):
                name = REF_SO
            super().__init__(name, *a, **k)

    ctypes.CDLL = _CDLL
    try:
        import matplotlib
        matplotlib.use("Agg")
        import gym  # the reference's cus_gym/gym
        from gym.wrappers import AssemblySwarmWrapper  # noqa: F401
        # gym.make imports the env module lazily; do it now while the redirect is active
        import gym.envs.customized_envs.assembly  # noqa: F401
    finally:
        ctypes.CDLL = real_cdll
    _gym = gym
    return gym


def ref_lib() -> ctypes.CDLL:
    """The compiled reference library itself (five C symbols)."""
    return ctypes.CDLL(REF_SO)


_results_file = None


def results_file() -> str:
    global _results_file
    if _results_file is None:
        sys.path.insert(0, REPO)
        from marl_amd.shapes import results_dict
        d = tempfile.mkdtemp(prefix="marl_amd_ref_")
        _results_file = os.path.join(d, "results.pkl")
        with open(_results_file, "wb") as f:
            pickle.dump(results_dict(), f)
    return _results_file


def default_args(**over) -> types.SimpleNamespace:
    """Defaults of marl_llm/cfg/assembly_cfg.py:159-179 (fields __reinit__ reads)."""
    a = dict(n_a=30, render_traj=False, traj_len=15, is_collected=False, is_boundary=True,
             dynamics_mode="Cartesian", agent_strategy="input", is_con_self_state=True,
             is_feature_norm=False, training_method="llm_rl", results_file=results_file())
    a.update(over)
    return types.SimpleNamespace(**a)


def make_ref_env(**over):
    """AssemblySwarmWrapper(gym.make('AssemblySwarm-v0').unwrapped, args) as
    marl_llm/train/train_assembly.py:55-58 builds it."""
    gym = _import_reference_gym()
    from gym.wrappers import AssemblySwarmWrapper
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        env = AssemblySwarmWrapper(gym.make("AssemblySwarm-v0").unwrapped, default_args(**over))
    return env
