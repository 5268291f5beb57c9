"""The allocation-free variant of _get_observation (asm_get_observation_noalloc, include/marl_amd_env.h) against
the existing entry point through the C ABI: bit-equal outputs on two calls in a row, the second with a smaller
input (the fixed device staging array keeps the first call's contents), and a clear error beyond its capacity."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BPOS = np.array([-2.4, 2.4, 2.4, -2.4])


class NoAlloc(ol.FiveFns):
    """FiveFns whose get_observation goes through asm_get_observation_noalloc and checks its status."""

    def __init__(self, cdll):
        super().__init__(cdll, False)
        fn = cdll.asm_get_observation_noalloc
        fn.restype = C.c_int
        self.status = []

        def f_obs(*a):
            rc = fn(*a)
            self.status.append(rc)
            assert rc == 0, cdll.asm_last_error()
        self.f_obs = f_obs


def _state(rng, n_a, shape):
    from marl_amd.shapes import load_shapes
    l_cells, grids = load_shapes()
    grid = np.ascontiguousarray(np.asarray(grids[shape], dtype=np.float64).T)
    pick = rng.choice(grid.shape[1], n_a, replace=False)
    p = np.ascontiguousarray(grid[:, pick] + 0.02 * rng.standard_normal((2, n_a)))      # inside: long lists
    dp = np.ascontiguousarray(rng.uniform(-0.5, 0.5, (2, n_a)))
    return p, dp, np.zeros_like(p), grid, float(l_cells[shape])


def test_noalloc_observation_matches_get_observation_twice():
    from marl_amd import _lib
    lib = _lib.load()
    assert lib.asm_device_count() >= 1, lib.asm_last_error()
    ref, new, orc = ol.FiveFns(lib, False), NoAlloc(lib), ol.oracle_fns()
    rng = np.random.default_rng(7)
    # second call smaller (fewer agents, fewer cells, shorter observation): stale bytes of the first must not show
    for n_a, shape, cond in ((64, 6, (False, True, True, False)), (12, 1, (False, True, False, False)),
                             (30, 3, (True, True, True, False))):
        p, dp, heading, grid, l_cell = _state(rng, n_a, shape)
        a = new.get_observation(p, dp, heading, BPOS, grid, 0.4, 0.26, l_cell, cond=cond)
        b = ref.get_observation(p, dp, heading, BPOS, grid, 0.4, 0.26, l_cell, cond=cond)
        c = orc.get_observation(p, dp, heading, BPOS, grid, 0.4, 0.26, l_cell, cond=cond)
        for name, x, y, z in zip(("obs", "nbr", "in_flags", "sensed", "occupied"), a, b, c):
            assert np.array_equal(x, y), (n_a, name)
            assert np.array_equal(x, z), (n_a, name, "oracle")
        assert (a[3] >= 0).sum() > n_a                       # the state exercises the lists
    assert new.status == [0, 0, 0]


def test_noalloc_rejects_oversized_request():
    from marl_amd import _lib
    lib = _lib.load()
    fn = lib.asm_get_observation_noalloc
    fn.restype = C.c_int
    n_a, n_g, n_obs, n_occ = 1024, 512, 400, 800            # 1024 * (400 + 800) * 4 B of lists alone > 4 MiB
    obs_dim = 32 + 2 * n_obs
    p = np.zeros((2, n_a)); grid = np.zeros((2, n_g)); obs = np.zeros((obs_dim, n_a))
    nbr = np.zeros((n_a, 6), np.int32); inf = np.zeros(n_a, np.int32)
    sen = np.zeros((n_a, n_obs), np.int32); occ = np.zeros((n_a, n_occ), np.int32)
    cond = np.array([False, True, True, False])
    rc = fn(ol.d(p), ol.d(p), ol.d(p), ol.d(obs), ol.d(BPOS), ol.d(grid), ol.i32(nbr), ol.i32(inf), ol.i32(sen), ol.i32(occ),
            C.c_double(0.4), C.c_double(0.05), C.c_double(0.06), C.c_double(0.8), C.c_int(6), C.c_int(n_obs), C.c_int(n_occ),
            C.c_int(n_a), C.c_int(n_g), C.c_int(obs_dim), C.c_int(2), ol.b8(cond))
    assert rc != 0 and b"fixed device array" in lib.asm_last_error()


def test_noalloc_calls_leave_device_memory_as_it_is():
    """The property the entry point exists for: after a warm-up call (the code object and the symbol look-up),
    further calls -- larger ones included -- do not change the device's free memory."""
    import torch
    from marl_amd import _lib
    lib = _lib.load()
    new = NoAlloc(lib)
    rng = np.random.default_rng(11)
    p, dp, heading, grid, l_cell = _state(rng, 8, 0)
    new.get_observation(p, dp, heading, BPOS, grid, 0.4, 0.26, l_cell)          # warm-up
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for n_a, shape in ((30, 2), (256, 6), (100, 4), (30, 1)):
        p, dp, heading, grid, l_cell = _state(rng, n_a, shape)
        new.get_observation(p, dp, heading, BPOS, grid, 0.4, 0.26, l_cell)
        torch.cuda.synchronize()
        free, _ = torch.cuda.mem_get_info()
        assert free == free0, (n_a, free0 - free)
    assert new.status == [0] * 5
