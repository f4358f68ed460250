"""solver=mg, the build's optional multigrid pressure solve ("better solver, non-default").
NOT the reference's arithmetic and with no counterpart there: its definition is oracle/cpu_ref_mg.h
(CR_MG), which the HIP implementation (csrc/multigrid.hip + the red-black pair kernel on level 0) must match
bit for bit; what ties the mode to the reference is that it solves the reference's own pressure equation
(the fixed point of linearSolver(0, p, div, 1, 6) + setBounds, simulation.cpp:263-271, 320), checked below."""
import numpy as np
import pytest

from conftest import ball_mask, bits_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def F():
    import fluid_simulation_amd as F
    return F


def tunnel_mask(W, H, D):
    m = ball_mask(W, H, D, W / 3.0, H / 2.0, D / 2.0, min(W, H, D) / 5.0)
    m[D // 4:3 * D // 4, H // 4:3 * H // 4, 2 * W // 3] = True      # a plate one cell thick
    m[1, 1, 1] = m[D, H, W] = True                                      # solids in two corners, next to three walls each
    m[0] = m[-1] = False
    m[:, 0] = m[:, -1] = False
    m[:, :, 0] = m[:, :, -1] = False
    return m


def relative_residual(p, div, mask):
    """|| div + sum of six neighbours - 6 p || over the fluid cells, relative to p = 0 (ghosts as stored)."""
    p = p.astype(np.float64)
    div = div.astype(np.float64)
    nb = p[1:-1, 1:-1, 2:] + p[1:-1, 1:-1, :-2] + p[1:-1, 2:, 1:-1] + p[1:-1, :-2, 1:-1] + p[2:, 1:-1, 1:-1] + p[:-2, 1:-1, 1:-1]
    r = div[1:-1, 1:-1, 1:-1] + nb - 6.0 * p[1:-1, 1:-1, 1:-1]
    live = ~mask[1:-1, 1:-1, 1:-1]
    return float(np.linalg.norm(r[live]) / np.linalg.norm(div[1:-1, 1:-1, 1:-1][live]))


@pytest.mark.parametrize("shape,fp64,mg", [((32, 32, 32), False, (3, 1, 1, 30)), ((64, 32, 16), False, (2, 2, 1, 10)),
                                           ((40, 24, 8), True, (2, 1, 2, 30)), ((128, 64, 64), False, (2, 1, 1, 30)),
                                           ((30, 20, 10), False, (2, 1, 1, 8)), ((33, 21, 5), False, (2, 1, 1, 4)),
                                           ((512, 16, 8), False, (1, 1, 1, 5)), ((24, 16, 16), True, (4, 1, 1, 30))])
def test_multigrid_steps_match_the_oracle_bit_for_bit(F, oracle_mod, shape, fp64, mg):
    """Whole steps under solver=mg against CR_MG: grids with 1 to 5 levels (33x21x5 cannot be halved at all: the
    cycles degenerate to red-black iterations on level 0), a plate one cell thick, corner solids, fp32 and fp64."""
    O = oracle_mod
    W, H, D = shape
    kw = dict(precision="fp64") if fp64 else {}
    sim = F.Simulation(W, H, D, 1, acc=5, solver="mg", quiet=1, mg_cycles=mg[0], mg_pre=mg[1], mg_post=mg[2],
                       mg_coarse_iters=mg[3], **kw)
    ora = O.Oracle(W, H, D, solver=O.MG, fp64=fp64, threads=8, acc=5, mg=mg)
    m = tunnel_mask(W, H, D)
    sim.set_mask(m)
    ora.set_mask(m)
    for _ in range(2):
        sim.run_one()
        ora.run_one()
    for f in range(11):
        assert bits_equal(sim.get(f), ora.get(f)), "%s %s: %s" % (shape, mg, F.FIELD_NAMES[f])
    levels = 1
    w, h, d = W, H, D
    while w % 2 == 0 and h % 2 == 0 and d % 2 == 0 and min(w, h, d) // 2 >= 4:
        w, h, d, levels = w // 2, h // 2, d // 2, levels + 1
    assert sim._geti("mg_levels") == levels


@pytest.mark.parametrize("shape", [(64, 64, 64), (128, 64, 64), (96, 48, 80)])
def test_multigrid_solves_the_reference_pressure_equation(F, shape):
    """The projection under solver=mg leaves the reference's pressure equation solved: after 8 V-cycles the
    residual of the fixed point of simulation.cpp:263-271 is below 1e-3 of its initial value on a developed flow around
    a ball and a plate (measured: 2e-4 and less), where 80 Jacobi sweeps leave more than a tenth (measured: 0.5)."""
    W, H, D = shape
    m = tunnel_mask(W, H, D)
    res = {}
    for solver, kw in (("mg", dict(mg_cycles=8)), ("jacobi", {})):
        sim = F.Simulation(W, H, D, 1, acc=80, solver=solver, quiet=1, **kw)
        sim.set_mask(m)
        for _ in range(3):
            sim.run_one()
        sim.project()
        res[solver] = relative_residual(sim.get(F.PRESSURE), sim.get(F.DIVERGENCE), m)
        sim.close()
    assert res["mg"] < 1e-3, res
    assert res["jacobi"] > 0.1, res


def test_multigrid_follows_obstacle_edits_and_odd_slabs_are_refused(F, oracle_mod):
    """The coarse operators are rebuilt when the obstacle field changes between steps.  On z-slabs the mode runs
    (tests/test_gpu_slabs.py::test_slabs_multigrid_matches_single_gpu) unless a rank would hold an odd number of planes:
    the eight children of a coarse cell must be one rank's."""
    O = oracle_mod
    W, H, D = 32, 16, 16
    sim = F.Simulation(W, H, D, 1, acc=4, solver="mg", quiet=1)
    ora = O.Oracle(W, H, D, solver=O.MG, threads=4, acc=4)
    for step in range(3):
        sim.addObstacle(5 + 6 * step, 8, 8)
        ora.add_obstacle(5 + 6 * step, 8, 8)
        sim.run_one()
        ora.run_one()
    for f in range(11):
        assert bits_equal(sim.get(f), ora.get(f)), F.FIELD_NAMES[f]
    slab = F.Simulation(W, H, 18, 1, acc=4, solver="mg", quiet=1)        # 9 planes per rank
    slab.comm_init(0, 2, b"FSNULL:".ljust(128, b"\0"))
    with pytest.raises(F.FluidsimError):
        slab.run_one()


def test_linear_solver_with_the_pressure_coefficients_runs_v_cycles(F, oracle_mod):
    """fs_linear_solver(0, x, x0, 1, 6) under solver=mg is the multigrid solve (on any pair of fields); other
    coefficients are relaxed as under jacobi.  Both against the oracle's cr_linear_solver."""
    O = oracle_mod
    W, H, D = 48, 32, 16
    m = tunnel_mask(W, H, D)
    sim = F.Simulation(W, H, D, 1, acc=6, solver="mg", quiet=1, mg_cycles=3)
    ora = O.Oracle(W, H, D, solver=O.MG, threads=4, acc=6, mg=(3, 1, 1, 30))
    rng = np.random.default_rng(11)
    rhs = rng.standard_normal((D + 2, H + 2, W + 2)).astype(np.float32)
    for x in (sim, ora):
        x.set_mask(m)
    sim.set(F.VY_PREV, rhs)
    ora.set(O.VY0, rhs)
    sim.linear_solver(0, F.BUFFER, F.VY_PREV, 1.0, 6.0)
    ora.linear_solver(0, O.BUF, O.VY0, 1.0, 6.0)
    assert bits_equal(sim.get(F.BUFFER), ora.get(O.BUF))
    assert relative_residual(sim.get(F.BUFFER), rhs, m) < 0.05
    sim.linear_solver(1, F.VX, F.VY_PREV, 0.3, 2.8)
    ora.linear_solver(1, O.VX, O.VY0, 0.3, 2.8)
    assert bits_equal(sim.get(F.VX), ora.get(O.VX))


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_multigrid_on_random_grids_and_random_solids(F, oracle_mod, seed):
    """Seeded random even grids with scattered solids (2-40 % of the cells: isolated fluid pockets, coarse cells that are
    entirely solid, fluid cells whose every neighbour is solid) and random cycle parameters, two steps, against the
    oracle bit for bit."""
    O = oracle_mod
    rng = np.random.default_rng(seed)
    W, H, D = (int(2 * rng.integers(4, 40)), int(2 * rng.integers(4, 24)), int(2 * rng.integers(4, 20)))
    if seed % 2 == 0:
        W, H, D = 8 * (W // 8 + 1), 8 * (H // 8 + 1), 8 * (D // 8 + 1)          # at least three levels
    fp64 = bool(seed % 3 == 0)
    mg = (int(rng.integers(1, 4)), int(rng.integers(1, 3)), int(rng.integers(1, 3)), int(rng.integers(1, 12)))
    m = np.zeros((D + 2, H + 2, W + 2), bool)
    m[1:-1, 1:-1, 1:-1] = rng.random((D, H, W)) < rng.choice([0.02, 0.1, 0.4])
    kw = dict(precision="fp64") if fp64 else {}
    sim = F.Simulation(W, H, D, 1, acc=3, solver="mg", quiet=1, mg_cycles=mg[0], mg_pre=mg[1], mg_post=mg[2],
                       mg_coarse_iters=mg[3], **kw)
    ora = O.Oracle(W, H, D, solver=O.MG, fp64=fp64, threads=8, acc=3, mg=mg)
    sim.set_mask(m)
    ora.set_mask(m)
    for _ in range(2):
        sim.run_one()
        ora.run_one()
    for f in range(11):
        assert bits_equal(sim.get(f), ora.get(f)), "%s %s seed %d: %s" % ((W, H, D), mg, seed, F.FIELD_NAMES[f])
