"""Edge cases of the C ABI on the GPU: degenerate parameters, obstacles edited between steps,
aliasing-sensitive call orders, repeated runs."""
import numpy as np
import pytest

from conftest import ball_mask, bits_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def F():
    import fluid_simulation_amd as F
    return F


def pair(F, O, W, H, D, solver, **kw):
    sim = F.Simulation(W, H, D, kw.get("iter", 1), solver=solver, quiet=1, **{k: v for k, v in kw.items() if k != "iter"})
    ora = O.Oracle(W, H, D, solver=O.JACOBI if solver == "jacobi" else O.GS_LEX, threads=1, **kw)
    return sim, ora


def same_state(F, O, sim, ora, what):
    for f in range(11):
        assert bits_equal(sim.get(f), ora.get(f)), "%s: %s" % (what, F.FIELD_NAMES[f])


@pytest.mark.parametrize("solver", ["jacobi", "gs_lex"])
def test_zero_solver_iterations(F, oracle_mod, solver):
    """acc = 0: no sweeps at all; the snapshots must still be real copies (advection reads them)."""
    O = oracle_mod
    sim, ora = pair(F, O, 14, 9, 7, solver, acc=0)
    m = ball_mask(14, 9, 7, 5, 4, 3, 2.2)
    sim.set_mask(m)
    ora.set_mask(m)
    for _ in range(3):
        sim.run_one()
        ora.run_one()
    same_state(F, O, sim, ora, "acc=0 " + solver)


def test_fp64_reference_order_mode(F, oracle_mod):
    """solver=gs_lex with fp64 fields against the fp64 oracle in GS_LEX mode."""
    O = oracle_mod
    W, H, D = 14, 10, 9
    sim = F.Simulation(W, H, D, 1, acc=4, solver="gs_lex", precision="fp64", quiet=1)
    ora = O.Oracle(W, H, D, solver=O.GS_LEX, fp64=True, threads=1, acc=4)
    m = ball_mask(W, H, D, 5, 5, 4, 2.0)
    sim.set_mask(m)
    ora.set_mask(m)
    for _ in range(2):
        sim.run_one()
        ora.run_one()
    same_state(F, O, sim, ora, "fp64 gs_lex")


@pytest.mark.parametrize("shape,fp64", [((50, 37, 41), False), ((70, 33, 17), True), ((33, 32, 33), False)])
def test_tiled_reference_order_sweep_on_ragged_grids(F, oracle_mod, shape, fp64):
    """Above 32768 cells solver=gs_lex sweeps 16^3 tiles in tile-hyperplane order, one launch per
    tile hyperplane; the result must still be the reference's lexicographic in-place sweep, also
    when the grid is not a multiple of the tile edge."""
    O = oracle_mod
    W, H, D = shape
    kw = dict(precision="fp64") if fp64 else {}
    sim = F.Simulation(W, H, D, 1, acc=3, solver="gs_lex", quiet=1, **kw)
    ora = O.Oracle(W, H, D, solver=O.GS_LEX, fp64=fp64, threads=1, acc=3)
    m = ball_mask(W, H, D, W // 3, H // 2, D // 2, min(H, D) / 4.0)
    sim.set_mask(m)
    ora.set_mask(m)
    for _ in range(2):
        sim.run_one()
        ora.run_one()
    same_state(F, O, sim, ora, "tiled gs_lex %s" % (shape,))


@pytest.mark.parametrize("shape,acc", [((14, 9, 7), 7), ((33, 21, 5), 8), ((300, 23, 9), 6), ((1, 1, 1), 3),
                                       ((256, 30, 14), 9), ((64, 40, 33), 10)])
def test_three_sweeps_per_pass_kernel_matches_oracle(F, oracle_mod, shape, acc):
    """sweep_fuse=4 runs every solve as passes of three fused sweeps (at the default, 3, the host
    only does so on grids where that times faster) plus a pair/single remainder; the result must be the oracle's Jacobi bit for bit,
    with solid cells in the corners next to all six walls (the ghost mirrors of zeroed cells)."""
    O = oracle_mod
    W, H, D = shape
    sim = F.Simulation(W, H, D, 1, acc=acc, solver="jacobi", quiet=1)
    sim.set_option("sweep_fuse", "4")
    try:
        ora = O.Oracle(W, H, D, solver=O.JACOBI, threads=4, acc=acc)
        m = ball_mask(W, H, D, W / 3.0, H / 2.0, D / 2.0, min(W, H, D) / 4.0)
        m[1, 1, 1] = m[D, H, W] = True
        m[D // 2 + 1, 1, W // 2 + 1] = True
        sim.set_mask(m)
        ora.set_mask(m)
        for _ in range(2):
            sim.run_one()
            ora.run_one()
        if W * H * D > 1:
            assert sim._geti("triple_plan") >= 0
        same_state(F, O, sim, ora, "three sweeps per pass %s" % (shape,))
    finally:
        sim.set_option("sweep_fuse", "3")       # the option is process-wide: back to the default


@pytest.mark.parametrize("shape,acc,fp64", [((24, 16, 12), 10, True), ((300, 20, 5), 4, True), ((512, 30, 9), 5, True),
                                            ((14, 9, 7), 7, True), ((1, 1, 1), 3, True), ((509, 23, 6), 3, True),
                                            ((600, 9, 7), 4, False), ((1000, 7, 6), 5, False), ((1024, 40, 12), 6, False),
                                            ((800, 30, 20), 7, False), ((520, 3, 2), 2, False), ((768, 11, 40), 4, False)])
def test_fused_two_sweep_kernel_matches_oracle(F, oracle_mod, shape, acc, fp64):
    """two_sweep_kernel=fused runs the two-sweep passes as jacobi_fused_kernel<NL=2> (fp64 rows up to 512
    cells; fp32 rows of 513..1024 cells) instead of leaving the choice to the clock: oracle Jacobi bit for
    bit, corner solids next to all six walls, ragged rows, odd iteration counts."""
    O = oracle_mod
    W, H, D = shape
    kw = dict(precision="fp64") if fp64 else {}
    sim = F.Simulation(W, H, D, 1, acc=acc, solver="jacobi", quiet=1, two_sweep_kernel="fused", **kw)
    ora = O.Oracle(W, H, D, solver=O.JACOBI, fp64=fp64, threads=4, acc=acc)
    m = ball_mask(W, H, D, W / 3.0, H / 2.0, D / 2.0, min(W, H, D) / 4.0)
    m[1, 1, 1] = m[D, H, W] = True
    m[D // 2 + 1, 1, W // 2 + 1] = True
    sim.set_mask(m)
    ora.set_mask(m)
    for _ in range(2):
        sim.run_one()
        ora.run_one()
    if acc >= 2:
        assert sim._geti("two_sweep_fused") == 1
    same_state(F, O, sim, ora, "fused two-sweep kernel %s" % (shape,))


def test_two_sweep_kernels_agree_at_benchmark_row_widths(F):
    """jacobi_pair_kernel against jacobi_fused_kernel<NL=2> on the GPU at the row widths of configs 4 and 5
    (lane-aligned variants): fp32 W = 1024, fp64 W = 512 and 256."""
    for (W, H, D, acc, prec) in [(1024, 45, 31, 8, "fp32"), (512, 45, 31, 6, "fp64"), (256, 70, 20, 6, "fp64")]:
        out = []
        for kind in ("fused", "pair"):
            sim = F.Simulation(W, H, D, 1, acc=acc, quiet=1, precision=prec, two_sweep_kernel=kind)
            m = ball_mask(W, H, D, W / 3.0, H / 2.0, D / 2.0, min(H, D) / 3.0)
            m[1, 1, 1] = m[D, H, W] = True
            sim.set_mask(m)
            sim.run_one()
            sim.run_one()
            assert sim._geti("two_sweep_fused") == (1 if kind == "fused" else 0)
            out.append([sim.get(f) for f in range(11)])
            sim.close()
        for f in range(11):
            assert bits_equal(out[0][f], out[1][f]), "%dx%dx%d %s %s" % (W, H, D, prec, F.FIELD_NAMES[f])


@pytest.mark.parametrize("speed,shape,fp64", [(30, (70, 33, 21), False), (1, (70, 33, 21), False), (-20, (66, 20, 17), False),
                                              (30, (300, 12, 9), True), (2, (23, 9, 40), True)])
def test_advection_row_kernels_match_cell_kernels_and_oracle(F, oracle_mod, speed, shape, fp64):
    """The default advection kernels (four cells per lane; traces whose x coordinate clamps read the
    pre-interpolated inlet / outlet column tables) against the per-cell kernels and the oracle: inlet speed 30
    (every trace clamps low), 1 (hardly any does), negative (clamps at the outlet side), ragged row ends."""
    O = oracle_mod
    W, H, D = shape
    kw = dict(precision="fp64") if fp64 else {}
    m = ball_mask(W, H, D, W / 3.0, H / 2.0, D / 2.0, min(W, H, D) / 4.0)
    m[1, 1, 1] = m[D, H, W] = True
    sims = [F.Simulation(W, H, D, 1, speed=speed, acc=4, quiet=1, advect_kernels=k, **kw) for k in ("row", "cell", "celltab")]
    # the tile kernels (inlet table windows staged in LDS): a window of 1 row / plane (most traces leave it and take the
    # per-cell path) and the default one
    sims += [F.Simulation(W, H, D, 1, speed=speed, acc=4, quiet=1, advect_kernels="tile", advect_window=w, **kw) for w in (1, 24)]
    ora = O.Oracle(W, H, D, solver=O.JACOBI, fp64=fp64, threads=4, speed=speed, acc=4)
    for x in sims + [ora]:
        x.set_mask(m)
    for _ in range(3):
        for x in sims + [ora]:
            x.run_one()
    for f in range(11):
        assert bits_equal(sims[0].get(f), sims[1].get(f)), "row vs cell: %s" % F.FIELD_NAMES[f]
        assert bits_equal(sims[2].get(f), sims[1].get(f)), "cell with clamp tables vs cell: %s" % F.FIELD_NAMES[f]
        assert bits_equal(sims[3].get(f), sims[1].get(f)), "tile kernels, window 1, vs cell: %s" % F.FIELD_NAMES[f]
        assert bits_equal(sims[4].get(f), sims[1].get(f)), "tile kernels vs cell: %s" % F.FIELD_NAMES[f]
    same_state(F, O, sims[0], ora, "advection row kernels, speed %d" % speed)
    for s_ in sims:
        s_.set_option("fuse_advect", "0")                # the three velocity advections as separate launches
    for x in sims + [ora]:
        x.run_one()
    same_state(F, O, sims[0], ora, "unfused row kernels, speed %d" % speed)
    same_state(F, O, sims[1], ora, "unfused cell kernels, speed %d" % speed)
    same_state(F, O, sims[2], ora, "unfused cell kernels with clamp tables, speed %d" % speed)
    same_state(F, O, sims[3], ora, "unfused tile kernels (window 1), speed %d" % speed)
    same_state(F, O, sims[4], ora, "unfused tile kernels, speed %d" % speed)


def test_three_sweeps_per_pass_kernel_full_rows(F):
    """The same against the pair kernel on the GPU at the row widths the benchmark grids use
    (W = 256 and 512 take the lane-aligned variant of the kernel)."""
    for (W, H, D, acc) in [(512, 45, 31, 11), (256, 70, 20, 7), (509, 12, 40, 10)]:
        out = []
        for fuse in ("4", "2"):
            sim = F.Simulation(W, H, D, 1, acc=acc, quiet=1)
            sim.set_option("sweep_fuse", fuse)
            m = ball_mask(W, H, D, W / 3.0, H / 2.0, D / 2.0, min(H, D) / 3.0)
            m[1, 1, 1] = m[D, H, W] = True
            sim.set_mask(m)
            sim.run_one()
            sim.run_one()
            out.append([sim.get(f) for f in range(11)])
            sim.set_option("sweep_fuse", "3")
            sim.close()
        for f in range(11):
            assert bits_equal(out[0][f], out[1][f]), "%dx%dx%d %s" % (W, H, D, F.FIELD_NAMES[f])


@pytest.mark.parametrize("shape,zc", [((512, 45, 40), 8), ((256, 70, 33), 5), ((256, 25, 24), 0), ((512, 10, 9), 3)])
def test_three_sweeps_wall_free_body_is_bit_identical(F, shape, zc):
    """wall_free=1: workgroups of the three-sweep kernel whose band and z chunk touch no wall run the second,
    wall-free body.  Forced here on grids with interior bands and (pair_zc) interior z chunks, with solids in the
    corners and in the middle, against the single general body; also a grid with no interior workgroup at all."""
    W, H, D = shape
    out = []
    for wf in ("1", "0"):
        sim = F.Simulation(W, H, D, 1, acc=10, quiet=1, wall_free=wf)
        sim.set_option("sweep_fuse", "4")
        if zc:
            sim.set_option("pair_zc", zc)
        m = ball_mask(W, H, D, W / 3.0, H / 2.0, D / 2.0, min(H, D) / 3.0)
        m[1, 1, 1] = m[D, H, W] = True
        sim.set_mask(m)
        sim.run_one()
        sim.run_one()
        out.append([sim.get(f) for f in range(11)])
        sim.set_option("sweep_fuse", "3")
        sim.close()
    for f in range(11):
        assert bits_equal(out[0][f], out[1][f]), "%dx%dx%d %s" % (W, H, D, F.FIELD_NAMES[f])


@pytest.mark.parametrize("shape,acc,omega,fp64", [((14, 9, 7), 5, 1.0, False), ((33, 21, 5), 4, 1.7, False),
                                                  ((300, 13, 9), 3, 1.5, False), ((1, 1, 1), 2, 1.9, False),
                                                  ((20, 12, 10), 4, 1.6, True)])
def test_red_black_sor_mode_matches_its_oracle(F, oracle_mod, shape, acc, omega, fp64):
    """solver=rbsor is the build's own optional solver (different arithmetic from the
    reference by design).  Its definition is oracle/cpu_ref.c CR_RBSOR; the GPU runs it as one pass of
    the pair kernel per iteration and must agree bit for bit, walls and corner solids included."""
    O = oracle_mod
    W, H, D = shape
    kw = dict(precision="fp64") if fp64 else {}
    sim = F.Simulation(W, H, D, 1, acc=acc, solver="rbsor", sor_omega=omega, quiet=1, **kw)
    ora = O.Oracle(W, H, D, solver=O.RBSOR, omega=omega, fp64=fp64, threads=4, acc=acc)
    m = ball_mask(W, H, D, W / 3.0, H / 2.0, D / 2.0, min(W, H, D) / 4.0)
    m[1, 1, 1] = m[D, H, W] = True
    sim.set_mask(m)
    ora.set_mask(m)
    for _ in range(2):
        sim.run_one()
        ora.run_one()
    same_state(F, O, sim, ora, "rbsor %s" % (shape,))


def test_red_black_sor_convergence_against_jacobi_at_equal_cost(F):
    """What the mode buys, measured as distance to the converged pressure (mean removed: the walls are
    all Neumann, only the obstacle pins the constant) on the pressure equation of a developed flow:
    with omega = 1 an iteration is worth the two Jacobi sweeps it costs; in the short fixed-`acc` regime
    of the reference the slow smooth modes dominate either way (omega 1.8: about 30 % less error after
    30 iterations); where a converged pressure is wanted, 1000 iterations at omega 1.9 end more than ten
    times closer than 2000 Jacobi sweeps."""
    W, H, D = 64, 48, 40
    mask = ball_mask(W, H, D, 20, 24, 20, 7.0)
    base = F.Simulation(W, H, D, 1, acc=8, quiet=1)
    base.set_mask(mask)
    for _ in range(5):
        base.run_one()
    div = base.get(F.DIVERGENCE)

    def solve(solver, n, **kw):
        sim = F.Simulation(W, H, D, 1, acc=n, quiet=1, solver=solver, **kw)
        sim.set_mask(mask)
        sim.set(F.DIVERGENCE, div)
        sim.linear_solver(0, F.PRESSURE, F.DIVERGENCE, 1.0, 6.0)
        return sim.get(F.PRESSURE).astype(np.float64)

    fluid = ~mask
    fluid[0] = fluid[-1] = False
    fluid[:, 0] = fluid[:, -1] = False
    fluid[:, :, 0] = fluid[:, :, -1] = False
    exact = solve("rbsor", 12000, sor_omega=1.9)

    def err(p):
        d = (p - exact)[fluid]
        d = d - d.mean()
        return float(np.sqrt(np.mean(d * d)))

    assert err(solve("rbsor", 6000, sor_omega=1.9)) < 1e-4 * err(np.zeros_like(exact))     # `exact` is converged
    e_gs, e_j = err(solve("rbsor", 30, sor_omega=1.0)), err(solve("jacobi", 60))
    assert abs(e_gs - e_j) < 0.05 * e_j, (e_gs, e_j)
    assert err(solve("rbsor", 30, sor_omega=1.8)) < 0.8 * e_j
    assert err(solve("rbsor", 1000, sor_omega=1.9)) < 0.1 * err(solve("jacobi", 2000))


def test_odd_and_single_iteration_counts(F, oracle_mod):
    O = oracle_mod
    for acc in (1, 2, 3, 7):
        sim, ora = pair(F, O, 18, 11, 9, "jacobi", acc=acc)
        for _ in range(2):
            sim.run_one()
            ora.run_one()
        same_state(F, O, sim, ora, "acc=%d" % acc)


def test_obstacles_edited_between_steps(F, oracle_mod):
    """The reference never edits obs after loading, but nothing forbids it: flags must follow."""
    O = oracle_mod
    sim, ora = pair(F, O, 16, 12, 10, "jacobi", acc=4)
    sim.run_one()
    ora.run_one()
    for s in (sim,):
        s.addObstacle(6, 6, 5)
        s.addObstacle(7, 6, 5)
    ora.add_obstacle(6, 6, 5)
    ora.add_obstacle(7, 6, 5)
    sim.run_one()
    ora.run_one()
    m = ball_mask(16, 12, 10, 11, 5, 5, 2.0)
    cur = ora.get(O.OBS) > 0.5
    sim.set_mask(m | cur)
    ora.set_mask(m | cur)
    sim.run_one()
    ora.run_one()
    same_state(F, O, sim, ora, "edited obstacles")


def test_mutators_between_steps_and_member_changes(F, oracle_mod):
    O = oracle_mod
    sim, ora = pair(F, O, 12, 10, 8, "jacobi", acc=5)
    sim.run_one()
    ora.run_one()
    sim.addDensity(4, 4, 4, 0.3)
    ora.add_density(4, 4, 4, 0.3)
    sim.setVelocity(5, 5, 5, 2.0, -1.0, 0.25)
    ora.set_velocity(5, 5, 5, 2.0, -1.0, 0.25)
    sim.step()           # step() alone: no inlet density, buffer keeps the previous dens
    ora.step_only()
    same_state(F, O, sim, ora, "mutators + step()")
    sim.speed = 12
    sim.acc = 3
    sim.dt = 0.02
    o2 = O.Oracle(12, 10, 8, solver=O.JACOBI, threads=1, acc=3, speed=12, dt=0.02)
    for f in range(11):
        o2.set(f, ora.get(f))
    sim.run_one()
    o2.run_one()
    same_state(F, O, sim, o2, "changed speed/acc/dt")


def test_set_get_roundtrip_and_types(F):
    sim = F.Simulation(9, 6, 5, 1, quiet=1)
    rng = np.random.default_rng(0)
    a = rng.standard_normal((7, 8, 11)).astype(np.float32)
    sim.set(F.VY, a)
    assert bits_equal(sim.get(F.VY), a)
    assert np.array_equal(sim.get(F.VY, dtype=np.float64), a.astype(np.float64))
    sim.set(F.DENS, a.astype(np.float64))
    assert bits_equal(sim.get(F.DENS), a)
    with pytest.raises(F.FluidsimError):
        sim.set(F.VX, a[:-1])                 # wrong size


def test_two_handles_are_independent(F, oracle_mod):
    O = oracle_mod
    s1, o1 = pair(F, O, 10, 8, 6, "jacobi", acc=4)
    s2, o2 = pair(F, O, 20, 6, 9, "jacobi", acc=2)
    for _ in range(2):
        s1.run_one()
        s2.run_one()
        o1.run_one()
        o2.run_one()
    same_state(F, O, s1, o1, "handle 1")
    same_state(F, O, s2, o2, "handle 2")


def test_run_twice_truncates_dumps(F, tmp_path):
    sim = F.Simulation(6, 5, 4, 2, acc=2, quiet=1, dump_dir=str(tmp_path))
    sim.run()
    sim.run()           # the reference's run() re-opens (truncates) its files
    sim.close()
    assert (tmp_path / "data.bin").stat().st_size == 2 * 8 * 7 * 6 * 4


def test_fs_run_statistics_lines(F, capfd):
    sim = F.Simulation(8, 6, 5, 100, acc=1, dump_every=0)
    sim.run()
    sim.close()
    out = capfd.readouterr().out
    assert "starting 3-D simulation: 8x6x5  steps = 100" in out
    assert "step 100" in out and "density sum = " in out
    for k in ("density  min", "density  max", "velocity x min", "velocity z max", "simulation finished"):
        assert k in out


def test_long_run_stays_bit_identical(F, oracle_mod):
    """300 steps of a developed flow around a ball (64x32x32, acc 15): no drift between the GPU and
    the oracle, checked every 100 steps."""
    O = oracle_mod
    W, H, D = 64, 32, 32
    sim, ora = pair(F, O, W, H, D, "jacobi", acc=15)
    m = ball_mask(W, H, D, 20, 16, 16, 6)
    sim.set_mask(m)
    ora.set_mask(m)
    for s in range(300):
        sim.run_one()
        ora.run_one()
        if (s + 1) % 100 == 0:
            same_state(F, O, sim, ora, "step %d" % (s + 1))
    assert float(sim.get(F.VX).max()) > 20.0


def test_create_destroy_does_not_leak_device_memory(F, tmp_path):
    """Handles own everything they allocate (arrays, flags, staging, writer buffers, events)."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so.7")
    free0, total = ctypes.c_size_t(), ctypes.c_size_t()

    def cycle(n):
        for _ in range(n):
            sim = F.Simulation(96, 64, 48, 2, acc=3, quiet=1, dump_dir=str(tmp_path))
            sim.addObstacle(10, 10, 10)
            sim.run()                      # steps + frame writer + statistics
            sim.get(F.VX)
            sim.close()

    cycle(3)                               # warm up allocator pools / code objects
    assert hip.hipMemGetInfo(ctypes.byref(free0), ctypes.byref(total)) == 0
    cycle(25)
    free1 = ctypes.c_size_t()
    assert hip.hipMemGetInfo(ctypes.byref(free1), ctypes.byref(total)) == 0
    leaked = int(free0.value) - int(free1.value)
    assert leaked < 64 << 20, "device memory shrank by %d MB over 25 create/run/destroy cycles" % (leaked >> 20)


def test_streamlines_match_the_viewer_restatement(F):
    """fs_streamlines against oracle/streamlines_ref.py (a statement-by-statement restatement of
    GUI/utils.py:40-213; parity unpinned: the reference module needs scikit-image and PyQt6 to import).
    Same seeds kept, same number of points per line, coordinates to 1e-9 -- the integration is float64
    on both sides and differs only in how numpy's BLAS dot and the device round the speed."""
    from oracle import streamlines_ref as R
    W, H, D = 40, 20, 18
    sim = F.Simulation(W, H, D, 1, acc=8, quiet=1)
    sim.set_mask(ball_mask(W, H, D, 14, 10, 9, 4.2))
    for _ in range(25):
        sim.run_one()
    vx, vy, vz, obs = (np.transpose(sim.get(f), (2, 1, 0)) for f in (F.VX, F.VY, F.VZ, F.OBS))   # main_window.py:227-230
    for density, thr in ((30, 0.1), (20, 0.0)):
        got, gnorm = sim.streamlines(density=density, vel_change_threshold=thr)
        want, wnorm = R.generate_streamlines(vx, vy, vz, obs, density=density, threshold=thr)
        assert len(got) == len(want) and len(want) > 0, (len(got), len(want))
        for a, b in zip(got, want):
            assert a.shape == b.shape
            assert np.allclose(a, b, rtol=0, atol=1e-9)
        assert np.allclose(gnorm, np.array(wnorm, dtype=np.float64), rtol=1e-12, atol=0)
    # the viewer-side drop-in takes the arrays the GUI holds (GUI/main_window.py:227-233)
    from fluid_simulation_amd.viewer import generate_streamlines
    lines2, colours = generate_streamlines(vx, vy, vz, obs, density=20, vel_change_threshold=0.0,
                                           cmap=lambda v: (v, 0.0, 1.0 - v, 1.0))
    assert len(lines2) == len(got) and all(np.array_equal(a, b) for a, b in zip(lines2, got))
    assert len(colours) == len(got) and colours[0].shape == (4,)
    # argument checks
    for bad in (dict(density=-1), dict(max_length=-2), dict(density=5000), dict(step_size=float("nan"))):
        with pytest.raises(F.FluidsimError):
            sim.streamlines(**bad)
    assert sim.streamlines(density=0) [0] == [] and sim.streamlines(max_length=0)[0] == []
    # no obstacles: the reference returns nothing
    empty = F.Simulation(12, 8, 6, 1, acc=2, quiet=1)
    empty.run_one()
    lines, norm = empty.streamlines()
    assert lines == [] and len(norm) == 0
