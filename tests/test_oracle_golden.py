"""The oracle (oracle/cpu_ref.c, GS_LEX mode, one thread) must reproduce, bit for bit,
every golden vector recorded from the compiled reference by oracle/make_golden.py.
This is what pins the oracle (the reference ships no tests of its own)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, ball_mask, bits_equal, load_golden, unpack_mask

G1 = ["g1_16c_empty_acc4", "g1_24x16x12_ball_acc20", "g1_12x10x8_wall_acc1", "g1_20x12x16_voxel_acc7",
      "g1_32c_ball_acc6"]


@pytest.mark.parametrize("name", G1)
def test_whole_steps_bit_exact(oracle_mod, name):
    O = oracle_mod
    meta, arr = load_golden(name)
    W, H, D = meta["W"], meta["H"], meta["D"]
    o = O.Oracle(W, H, D, solver=O.GS_LEX, threads=1, iter=meta["steps"], acc=meta["acc"])
    o.set_mask(unpack_mask(arr["mask"], W, H, D))
    checked = 0
    for s in range(1, meta["steps"] + 1):
        o.run_one()
        for f, fname in enumerate(O.FIELD_NAMES):
            key = "s%d_%s" % (s, fname)
            if key in arr:
                assert bits_equal(o.get(f), arr[key]), key
                checked += 1
    assert checked >= 6


def _state_oracle(O, meta, arr):
    W, H, D = meta["W"], meta["H"], meta["D"]
    o = O.Oracle(W, H, D, solver=O.GS_LEX, threads=1, acc=meta["acc"])
    o.set_mask(unpack_mask(arr["mask"], W, H, D))
    for f, fname in enumerate(O.FIELD_NAMES):
        if fname != "obs":
            o.set(f, arr["in_" + fname])
    return o


def test_single_passes_bit_exact(oracle_mod):
    O = oracle_mod
    meta, arr = load_golden("g2_passes_24x16x12")
    for b in (0, 1, 2, 3):
        o = _state_oracle(O, meta, arr)
        o.set_bounds(b, O.VX)
        assert bits_equal(o.get(O.VX), arr["set_bounds_b%d_v_x" % b])
    for b, fld, prv in ((1, O.VX, O.VX0), (2, O.VY, O.VY0), (3, O.VZ, O.VZ0), (0, O.DENS, O.BUF)):
        o = _state_oracle(O, meta, arr)
        o.diffuse(b, fld, prv)
        assert bits_equal(o.get(fld), arr["diffuse_b%d" % b])
        o = _state_oracle(O, meta, arr)
        o.advect(b, fld, prv)
        assert bits_equal(o.get(fld), arr["advect_b%d" % b])
    o = _state_oracle(O, meta, arr)
    o.linear_solver(0, O.P, O.DIV, 1.0, 6.0)
    assert bits_equal(o.get(O.P), arr["linear_solver_p"])
    o = _state_oracle(O, meta, arr)
    o.project()
    for f in (O.VX, O.VY, O.VZ, O.P, O.DIV):
        assert bits_equal(o.get(f), arr["project_" + O.FIELD_NAMES[f]])


@pytest.mark.parametrize("name", ["g3_sphere_32x24x20", "g3_sphere_48c_big", "g3_plate_rot_32x24x20"])
def test_voxelizer_mask_bit_exact(oracle_mod, name):
    O = oracle_mod
    meta, arr = load_golden(name)
    W, H, D = meta["W"], meta["H"], meta["D"]
    o = O.Oracle(W, H, D)
    n = o.load_stl(os.path.join(GOLDEN, meta["stl"]), scale=meta["scale"], rot=meta["rot"],
                   translate=meta["translate"], seed=meta["seed"])
    assert n > 0
    got = o.get(O.OBS) > 0.5
    want = unpack_mask(arr["mask"], W, H, D)
    assert int(got.sum()) == meta["solids"]
    assert np.array_equal(got, want)


def test_voxelizer_two_meshes(oracle_mod):
    O = oracle_mod
    meta, arr = load_golden("g3_sphere_plus_plate_40x24x24")
    W, H, D = meta["W"], meta["H"], meta["D"]
    o = O.Oracle(W, H, D)
    o.load_stl(os.path.join(GOLDEN, "sphere_24x12.stl"), scale=0.4, translate=(-8.0, 0.0, 0.0), seed=meta["seed"])
    o.load_stl(os.path.join(GOLDEN, "plate_ascii.stl"), scale=0.7, translate=(6.0, 0.0, 0.0), seed=meta["seed"])
    assert np.array_equal(o.get(O.OBS) > 0.5, unpack_mask(arr["mask"], W, H, D))


def test_voxelizer_missing_file_leaves_tunnel_empty(oracle_mod):
    O = oracle_mod
    o = O.Oracle(8, 8, 8)
    assert o.load_stl("/nonexistent/none.stl") == -1
    assert not o.get(O.OBS).any()


def test_dump_layout_byte_exact(oracle_mod, tmp_path):
    """simulation.cpp:140-148 + the viewers' assumptions (GUI/main_window.py:159-172,
    gui.py:228-231): file size is an exact multiple of the padded frame, frames are
    C-order (T, D+2, H+2, W+2) float32."""
    O = oracle_mod
    meta, arr = load_golden("g4_layout_8x6x4")
    W, H, D = meta["W"], meta["H"], meta["D"]
    o = O.Oracle(W, H, D, solver=O.GS_LEX, threads=1, iter=meta["steps"], acc=meta["acc"])
    for x, y, z in meta["obstacles"]:
        o.add_obstacle(x, y, z)
    for s in range(meta["steps"]):
        o.run_one()
        o.dump_frame(str(tmp_path), append=(s > 0))
    frame = (W + 2) * (H + 2) * (D + 2) * 4
    for fn in ("data", "obs", "v_x", "v_y", "v_z"):
        got = np.fromfile(str(tmp_path / (fn + ".bin")), dtype=np.uint8)
        assert got.size == meta["steps"] * frame
        assert np.array_equal(got, arr[fn]), fn
    obs = arr["obs"].view(np.float32).reshape(meta["steps"], D + 2, H + 2, W + 2)
    assert obs[-1, 2, 3, 3] == 1.0 and obs[-1, 2, 3, 4] == 1.0 and (obs > 0.5).sum() == 2 * meta["steps"]


def test_jacobi_differs_from_gs_but_is_thread_independent(oracle_mod):
    """Jacobi is the north-star solver; it is a different iteration from the reference's
    in-place sweep (so results differ at O(1e-1)) but, unlike it, does not depend on the
    thread count."""
    O = oracle_mod
    W, H, D = 16, 12, 10
    res = []
    for threads in (1, 4):
        o = O.Oracle(W, H, D, solver=O.JACOBI, threads=threads, acc=8)
        for _ in range(2):
            o.run_one()
        res.append(o.get(O.VX))
    assert bits_equal(res[0], res[1])
    g = O.Oracle(W, H, D, solver=O.GS_LEX, threads=1, acc=8)
    for _ in range(2):
        g.run_one()
    assert not bits_equal(res[0], g.get(O.VX))


def test_stock_run_first_40_frames_digest(oracle_mod, tmp_path):
    """The reference program's default run (simulation.cpp:429-451; empty tunnel because its
    hard-coded STL path does not exist), reduced to SHA-256 digests by oracle/make_golden.py.
    The oracle replays the first 40 of the 100 steps and must produce the same bytes on disk."""
    import hashlib
    import json
    O = oracle_mod
    meta = json.load(open(os.path.join(GOLDEN, "g5_stock_run_digests.json")))
    W, H, D = meta["W"], meta["H"], meta["D"]
    o = O.Oracle(W, H, D, solver=O.GS_LEX, threads=1, iter=40, acc=meta["acc"], speed=meta["speed"])
    hashes = {fn: hashlib.sha256() for fn in meta["files"]}
    for s in range(40):
        o.run_one()
        o.dump_frame(str(tmp_path), append=False)          # one frame at a time keeps the disk use small
        for fn, h in hashes.items():
            h.update((tmp_path / (fn + ".bin")).read_bytes())
    for fn, h in hashes.items():
        assert h.hexdigest() == meta["files"][fn]["sha256_first_40_frames"], fn


def test_red_black_sor_oracle_is_thread_independent(oracle_mod):
    """CR_RBSOR is the build's own optional solver, not a restatement of reference code (no goldens can
    exist); what the CPU suite can pin is that its in-place colour loops have no ordering freedom: one
    thread and four threads give the same bits, and omega = 1 after many iterations solves the same
    equation as Jacobi (same fixed point)."""
    O = oracle_mod
    W, H, D = 13, 9, 7
    m = ball_mask(W, H, D, 5, 4, 3, 2.0)
    runs = []
    for threads in (1, 4):
        o = O.Oracle(W, H, D, solver=O.RBSOR, omega=1.7, threads=threads, acc=6)
        o.set_mask(m)
        for _ in range(3):
            o.run_one()
        runs.append([o.get(f) for f in range(11)])
    for a, b in zip(*runs):
        assert bits_equal(a, b)
    # same fixed point as Jacobi: solve one diffusion equation to convergence both ways
    rng = np.random.default_rng(5)
    x0 = rng.standard_normal((D + 2, H + 2, W + 2)).astype(np.float32)
    sols = []
    for solver, acc, kw in ((O.JACOBI, 400, {}), (O.RBSOR, 200, {"omega": 1.0})):
        o = O.Oracle(W, H, D, solver=solver, threads=2, acc=acc, **kw)
        o.set(O.VY0, x0)
        o.linear_solver(0, O.VY, O.VY0, 0.8, 1 + 6 * 0.8)
        sols.append(o.get(O.VY)[1:-1, 1:-1, 1:-1])
    assert np.allclose(sols[0], sols[1], rtol=0, atol=2e-6)
