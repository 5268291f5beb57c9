"""CPU-side checks: the C-ABI library exists, loads and exports every declared
symbol (no compute calls here), header and binding agree, the gym-shaped shim and the
shape pipeline behave like the reference's."""
import ctypes
import os
import re

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    from marl_amd import _lib
    assert os.path.isfile(_lib.SO_PATH), "build with `make -C marl_amd/csrc`"
    lib = ctypes.CDLL(_lib.SO_PATH)
    header = open(os.path.join(REPO, "include", "marl_amd_env.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(_get_observation|_get_reward|_sf_b2b_all|_get_dist_b2w|calculateActionPrior|asm_\w+)\s*\(", header))
    declared.discard("asm_params")
    assert declared == set(_lib.EXPORTS)
    for sym in declared:
        assert getattr(lib, sym) is not None
    lib.asm_abi_version.restype = ctypes.c_int
    assert lib.asm_abi_version() == _lib.ABI_VERSION        # pure host call


def test_params_struct_matches_header_defaults():
    from marl_amd import _lib
    p = _lib.default_params(30, 544, 0.26)
    assert (p.n_a, p.n_g_max, p.topo_nei_max, p.num_obs_grid_max, p.num_occupied_grid_max) == (30, 544, 6, 80, 200)
    assert (p.size_a, p.d_sen, p.r_avoid, p.vel_max, p.k_ball, p.k_wall, p.c_wall, p.dt) == (0.035, 0.4, 0.26, 0.8, 30, 100, 5, 0.1)
    assert _lib.load().asm_obs_dim(ctypes.byref(p)) == 192   # ENV:806
    p.is_con_self_state = 0
    assert _lib.load().asm_obs_dim(ctypes.byref(p)) == 188


def test_bad_argument_to_a_reference_symbol_returns_instead_of_aborting():
    """No GPU needed: the argument check comes before any HIP call.  The outputs are marked, the status is set."""
    import ctypes as C
    from marl_amd import _lib
    lib = _lib.load()
    n_a, n_g = 4, 8
    p = np.zeros((2, n_a)); obs = np.zeros((192, n_a)); grid = np.zeros((2, n_g)); bpos = np.array([-2.4, 2.4, 2.4, -2.4])
    nbr = np.zeros((n_a, 5), np.int32); inf = np.zeros(n_a, np.int32); sen = np.zeros((n_a, 80), np.int32); occ = np.zeros((n_a, 200), np.int32)
    cond = np.array([False, True, True, False])
    D = lambda a: a.ctypes.data_as(C.POINTER(C.c_double)); I = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    lib._get_observation(D(p), D(p), D(p), D(obs), D(bpos), D(grid), I(nbr), I(inf), I(sen), I(occ), C.c_double(0.4),
                         C.c_double(0.26), C.c_double(0.06), C.c_double(0.8), C.c_int(5), C.c_int(80), C.c_int(200),
                         C.c_int(n_a), C.c_int(n_g), C.c_int(192), C.c_int(2), cond.ctypes.data_as(C.POINTER(C.c_bool)))
    assert lib.asm_compat_status() != 0 and b"topo_nei_max" in lib.asm_last_error()
    assert np.isnan(obs).all() and (nbr == -1).all() and (sen == -1).all() and (occ == -1).all() and (inf == -1).all()
    with pytest.raises(RuntimeError, match="topo_nei_max"):
        _lib.check_compat("_get_observation")


def test_missing_library_fails_loudly(monkeypatch):
    from marl_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "SO_PATH", "/nonexistent/libAssemblyEnv.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


def test_batched_env_refuses_cpu_device():
    from marl_amd.batched import BatchedAssemblyEnv
    with pytest.raises(RuntimeError, match="HIP device only"):
        BatchedAssemblyEnv(2, 30, device="cpu")


def test_shapes_match_reference_table():
    """n_g and l_cell of the seven shapes (survey of the reference's fig/*.png)."""
    from marl_amd.shapes import load_shapes, otsu_threshold, grid_from_gray
    l_cells, grids = load_shapes()
    assert [g.shape[0] for g in grids] == [497, 487, 534, 516, 510, 502, 536]
    np.testing.assert_allclose(l_cells, [0.061111, 0.057895, 0.066667, 0.064706, 0.062857, 0.070968, 0.062857], atol=5e-7)
    for g in grids:
        assert abs(g[:, 0].mean()) < 1e-12 and abs(g[:, 1].mean()) < 1e-12
        assert g[:, 1].max() - g[:, 1].min() == pytest.approx(2.2)
    # the pipeline itself on a synthetic image: a black 10 x 6 cell rectangle on white
    img = np.full((36 * 10, 36 * 14), 255, np.uint8)
    img[36:36 * 9, 36 * 2:36 * 12] = 0
    assert otsu_threshold(img) < 255
    l_cell, grid = grid_from_gray(img)
    assert grid.shape == (6 * 8, 2)                      # the outermost ring of cells is skipped (cfg:64-65)
    assert l_cell == pytest.approx(2.2 / 5)


REF_FIG = "/root/reference/fig"


@pytest.mark.skipif(not os.path.isdir(REF_FIG), reason="needs the reference's fig/*.png (only in the build container)")
def test_packaged_grids_equal_the_pipeline_on_the_reference_images():
    """marl_amd/data/shapes.npz is exactly what the PNG pipeline (assembly_cfg.py:47-102 restated in
    marl_amd/shapes.py) yields on the reference's seven images."""
    from marl_amd.shapes import grid_from_png, load_shapes, sorted_pngs
    l_cells, grids = load_shapes()
    paths = sorted_pngs(REF_FIG)
    assert [os.path.basename(p) for p in paths] == ["1.png", "3.png", "4.png", "5.png", "6.png", "7.png", "8.png"]
    for k, path in enumerate(paths):
        l_cell, grid = grid_from_png(path)
        assert l_cell == l_cells[k] and np.array_equal(grid, grids[k]), path


def test_gym_shim_semantics():
    import marl_amd.gym as gym
    from marl_amd.gym.core import TimeLimit
    env = gym.make("AssemblySwarm-v0")
    assert isinstance(env, TimeLimit) and env.spec.max_episode_steps == 100   # registration.py:105-110
    base = env.unwrapped
    assert type(base).__name__ == "AssemblySwarmEnv" and base.unwrapped is base
    assert (base.n_a, base.topo_nei_max, base.dt, base.k_ball, base.k_wall, base.c_wall) == (10, 6, 0.1, 30, 100, 5)
    w = gym.Wrapper(base)
    assert w.n_a == 10                                   # core.py:232-237 attribute forwarding
    with pytest.raises(AttributeError):
        w._private
    w.env.alpha = 0.1
    assert base.alpha == 0.1
    box = gym.spaces.Box(-np.inf, np.inf, shape=(192, 30), dtype=np.float32)
    assert box.shape == (192, 30) and box.dtype == np.float32


def test_compat_import_gym_alias():
    import subprocess, sys
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import gym; "
            "from gym.wrappers import AssemblySwarmWrapper; from gym import spaces; "
            "e = gym.make('AssemblySwarm-v0').unwrapped; print(type(e).__name__, AssemblySwarmWrapper.__name__)"
            % (os.path.join(REPO, "marl_amd", "compat"), REPO))
    out = subprocess.check_output([sys.executable, "-c", code], text=True)
    assert out.split() == ["AssemblySwarmEnv", "AssemblySwarmWrapper"]


def test_wrapper_metrics_match_reference_values():
    """Vectorised metrics against values recorded from the reference wrapper."""
    from marl_amd.gym.wrappers import AssemblySwarmWrapper
    z = np.load(os.path.join(REPO, "tests", "golden", "metrics.npz"))
    for k in range(int(z["count"])):
        w = AssemblySwarmWrapper.__new__(AssemblySwarmWrapper)
        class E:  # the attributes the metrics read
            pass
        e = E(); e.p = z[f"p_{k}"]; e.grid_center = z[f"grid_{k}"]; e.n_g = e.grid_center.shape[1]; e.n_a = e.p.shape[1]
        e.r_avoid = float(z[f"r_avoid_{k}"]); e.is_collide_b2b = z[f"collide_{k}"]
        w.env = e
        m = w.get_formation_metrics()
        assert m["coverage_rate"] == float(z[f"coverage_rate_{k}"])
        assert m["num_collisions"] == int(z[f"num_collisions_{k}"])
        for key, zk in (("distribution_uniformity",) * 2, ("voronoi_uniformity",) * 2, ("avg_distance_to_target",) * 2):
            assert m[key] == pytest.approx(float(z[f"{zk}_{k}"]), rel=1e-12, abs=1e-14)


def test_bench_roofline_inputs():
    """The bench's roofline figures: algorithmic bytes and FLOPs per agent-step, the committed counter summaries it
    quotes the measured traffic from (only for the workload the counters were taken on), and the committed bench
    lines' internal consistency: frac == value per GPU x algorithmic bytes (FLOPs) / peak, bound chosen by intensity."""
    import json
    import bench
    assert bench.b_alg_per_agent_step(30, 500.0) == pytest.approx(872 + 8 * 500 / 30)
    assert bench.f_alg_per_agent_step(1024, 500.0) == pytest.approx(15 * 500 + 20 * 1023 + 2400)
    t = bench.measured_traffic(4096, 30, 3)                  # (bytes per launch: the counters belong to one shard size)
    assert t is not None and t["source"] == "profiles/r03_pmc.json"
    assert t["hbm_bytes_per_launch"] > 0.9 * 1365 * 30 * bench.b_alg_per_agent_step(30, 511.8)
    # (a run sharded otherwise quotes the same file: bytes per launch x the file's launches per step are a step's bytes)
    assert bench.measured_traffic(4096, 30, 2)["workload"]["streams"] == 3
    assert bench.measured_traffic(4096, 31, 3) is None
    assert bench.measured_traffic(256, 1024, 1)["source"] == "profiles/r03_c4_pmc.json"
    assert bench.measured_traffic(256, 1024, 4)["source"] == "profiles/r03_c4_pmc.json"      # counter passes serialise the launches
    assert bench.profiled_kernel(256, 1024, 4)["kernel"] == "asmenv::k_step<float, true>"
    assert bench.step_kernel_name(32 + 64 + 256, False, 0, False).startswith("asmenv::k_step<float, true>")
    assert bench.step_kernel_name(4 + 16, False, 0, False) == "asmenv::k_step_fast<4, false, 0, true, false>"
    lines = [json.load(open(os.path.join(REPO, "profiles", "r03_bench_default.json"))),
             json.load(open(os.path.join(REPO, "profiles", "r03_bench_steps20_warmup5.json")))]
    lines += [json.loads(l) for l in open(os.path.join(REPO, "profiles", "r03_bench_variants.jsonl"))]
    for line in lines:
        r, c = line["roofline"], line["config"]
        per_gpu = line["value"] / line["n_gpus"]
        if c["agents"] == 1024:
            assert r["bound"] == "valu" and r["peak"] == 157.3 and r["unit"] == "TFLOP/s"
            assert r["achieved"] == pytest.approx(per_gpu * r["alg_flops_per_agent_step"] / 1e12)
        else:
            assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
            assert r["achieved"] == pytest.approx(per_gpu * r["alg_bytes_per_agent_step"] / 1e9)
        assert r["frac"] == pytest.approx(r["achieved"] / r["peak"])
        if r["traffic"] is not None:                          # counters' bytes of a whole step / this line's step time
            step_bytes = r["traffic_bytes_per_launch"] * r["traffic_launches_per_step"]
            assert r["traffic"] == pytest.approx(step_bytes / (line["ms_per_step"] * 1e-3) / 1e9)
            assert 0.9 * c["envs_per_gpu"] * c["agents"] * r["alg_bytes_per_agent_step"] < step_bytes
            assert r["traffic"] < 8000.0
        if r["kernel_avg_us_profiled"] is not None:           # quoted only for a run sharded like the traced one
            assert r["kernel_avg_us_profiled"] < r["avg_launch_ms"] * 1e3 and r["launch_gap_us"] > 0
        assert line["value"] == pytest.approx(c["global_envs"] * c["agents"] / (line["ms_per_step"] * 1e-3))
        assert line["timing"]["windows"] >= 5 and line["timing"]["steps_per_window"] == line["steps"]
        assert f'{c["agents"]}-agent' in line["metric"]
    # a 20-step run measures the same step time as the 2000-step run (both taken on one box, one after the other)
    assert lines[1]["ms_per_step"] == pytest.approx(lines[0]["ms_per_step"], rel=0.05)
    assert lines[0]["cpu_baseline"]["kind"] in ("reference", "port") and lines[0]["cpu_baseline"]["cores"] >= 1


def test_lattice_tables_of_the_reference_shapes():
    """asm_lattice_build (host code of the library): every cell is found again through the row windows, the
    boundary list holds exactly the cells with a missing 4-neighbour, and non-lattice input is refused."""
    from marl_amd import _lib
    from marl_amd.shapes import load_shapes
    lib = _lib.load()
    l_cells, grids = load_shapes()
    for l_cell, g in zip(l_cells, grids):
        g = np.ascontiguousarray(g, dtype=np.float64)
        sh = _lib.AsmLatticeShape()
        assert lib.asm_lattice_build(g.ctypes.data, g.shape[0], l_cell, ctypes.byref(sh)) == 0, lib.asm_last_error()
        col = np.rint((g[:, 0] - g[:, 0].min()) / l_cell).astype(int); row = np.rint((g[:, 1] - g[:, 1].min()) / l_cell).astype(int)
        assert (sh.cols, sh.rows, sh.n_g) == (col.max() + 1, row.max() + 1, g.shape[0])
        assert sh.x0 == g[:, 0].min() and sh.y0 == g[:, 1].min() and sh.l_cell == l_cell
        tab = np.ctypeslib.as_array(sh.row_tab)                      # [48][3][2]
        sites = {(int(a), int(b)): c for c, (a, b) in enumerate(zip(col, row))}
        for (a, b), c in sites.items():
            for j in range(3):
                if 16 * j <= a < 16 * j + 32:
                    bits, first = int(tab[b + 1, j, 0]), int(tab[b + 1, j, 1])
                    assert (bits >> (a - 16 * j)) & 1
                    assert first + bin(bits & ((1 << (a - 16 * j)) - 1)).count("1") == c
        assert sum(bin(int(tab[r, 0, 0]) & 0xFFFF).count("1") + bin(int(tab[r, 1, 0]) & 0xFFFF).count("1") + bin(int(tab[r, 2, 0])).count("1")
                   for r in range(48)) == g.shape[0]
        assert not tab[0].any() or int(tab[0, :, 0].sum()) == 0      # the row below the lattice is empty
        want = [c for (a, b), c in sorted(sites.items(), key=lambda kv: kv[1])
                if not all(n in sites for n in ((a + 1, b), (a - 1, b), (a, b + 1), (a, b - 1)))]
        got = list(np.ctypeslib.as_array(sh.boundary)[:sh.n_boundary])
        assert got == want and 100 < len(got) <= 256
    g = np.ascontiguousarray(grids[0], dtype=np.float64).copy()
    g[5, 0] += 0.3 * l_cells[0]
    assert lib.asm_lattice_build(g.ctypes.data, g.shape[0], l_cells[0], ctypes.byref(_lib.AsmLatticeShape())) != 0
    g = np.ascontiguousarray(grids[0][::-1], dtype=np.float64)        # lattice, but not in row-major order
    assert lib.asm_lattice_build(g.ctypes.data, g.shape[0], l_cells[0], ctypes.byref(_lib.AsmLatticeShape())) != 0
    assert b"order" in lib.asm_last_error()
