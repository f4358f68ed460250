"""Randomised parity: hypothesis draws grid shapes, parameters, obstacle sets and call sequences;
the GPU (through the C ABI) must stay bit-identical with the oracle in the same solver mode."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from conftest import bits_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods(oracle_mod):
    import fluid_simulation_amd as F
    return F, oracle_mod


dims = st.tuples(st.integers(1, 70), st.integers(1, 40), st.integers(1, 30))


@settings(max_examples=30, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture], derandomize=True)
@given(dims=dims, acc=st.integers(0, 9), speed=st.integers(0, 45), dt=st.sampled_from([0.05, 0.02, 0.1]),
       diff=st.sampled_from([2.0e-5, 1.0e-3, 0.0]), solver=st.sampled_from(["jacobi", "jacobi", "gs_lex"]),
       seed=st.integers(0, 2 ** 31 - 1), steps=st.integers(1, 3))
def test_random_runs_match_oracle(mods, dims, acc, speed, dt, diff, solver, seed, steps):
    F, O = mods
    W, H, D = dims
    rng = np.random.default_rng(seed)
    kw = dict(iter=steps, acc=acc, speed=speed, dt=dt, diff=diff)
    sim = F.Simulation(W, H, D, steps, speed, dt, diff, 1.5e-5, acc, solver=solver, quiet=1)
    ora = O.Oracle(W, H, D, solver=O.JACOBI if solver == "jacobi" else O.GS_LEX, **kw)
    mask = np.zeros((D + 2, H + 2, W + 2), dtype=bool)
    mask[1:-1, 1:-1, 1:-1] = rng.random((D, H, W)) < rng.choice([0.0, 0.03, 0.2])
    sim.set_mask(mask)
    ora.set_mask(mask)
    for _ in range(int(rng.integers(0, 4))):
        x, y, z = int(rng.integers(1, W + 1)), int(rng.integers(1, H + 1)), int(rng.integers(1, D + 1))
        amt = float(np.float32(rng.uniform(-1, 1)))
        sim.addDensity(x, y, z, amt)
        ora.add_density(x, y, z, amt)
        sim.setVelocity(x, y, z, amt, -amt, 0.5 * amt)
        ora.set_velocity(x, y, z, amt, -amt, 0.5 * amt)
    for s in range(steps):
        if rng.random() < 0.3:
            sim.step()
            ora.step_only()
        else:
            sim.run_one()
            ora.run_one()
    for f in range(11):
        assert bits_equal(sim.get(f), ora.get(f)), (dims, acc, solver, F.FIELD_NAMES[f])
    sim.close()


@settings(max_examples=8, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture], derandomize=True)
@given(dims=st.tuples(st.integers(2, 300), st.integers(2, 24), st.integers(2, 12)), acc=st.integers(1, 6),
       seed=st.integers(0, 2 ** 31 - 1))
def test_random_fp64_runs_match_oracle(mods, dims, acc, seed):
    F, O = mods
    W, H, D = dims
    rng = np.random.default_rng(seed)
    sim = F.Simulation(W, H, D, 1, acc=acc, precision="fp64", quiet=1)
    ora = O.Oracle(W, H, D, solver=O.JACOBI, fp64=True, acc=acc)
    mask = np.zeros((D + 2, H + 2, W + 2), dtype=bool)
    mask[1:-1, 1:-1, 1:-1] = rng.random((D, H, W)) < 0.05
    sim.set_mask(mask)
    ora.set_mask(mask)
    for _ in range(2):
        sim.run_one()
        ora.run_one()
    for f in range(11):
        assert bits_equal(sim.get(f), ora.get(f)), (dims, acc, F.FIELD_NAMES[f])
    sim.close()


@settings(max_examples=24, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture], derandomize=True)
@given(dims=st.tuples(st.one_of(st.integers(1, 40), st.integers(250, 330)), st.one_of(st.integers(1, 12), st.integers(20, 40)),
                     st.integers(1, 28)), acc=st.integers(1, 10),
       mode=st.sampled_from(["triple", "rbsor"]), omega=st.sampled_from([0.7, 1.0, 1.5, 1.9]),
       seed=st.integers(0, 2 ** 31 - 1))
def test_random_runs_of_the_optional_kernels_match_oracle(mods, dims, acc, mode, omega, seed):
    """Two solver paths the small default-run grids rarely take, under random grids and obstacle sets:
    the three-sweeps-per-pass kernel forced on (sweep_fuse=4; must equal plain Jacobi) and the red-black SOR mode
    (must equal its oracle definition), random solids including cells on the walls."""
    F, O = mods
    W, H, D = dims
    rng = np.random.default_rng(seed)
    if mode == "triple":
        sim = F.Simulation(W, H, D, 1, acc=acc, quiet=1)
        sim.set_option("sweep_fuse", "4")
        ora = O.Oracle(W, H, D, solver=O.JACOBI, acc=acc)
    else:
        sim = F.Simulation(W, H, D, 1, acc=acc, solver="rbsor", sor_omega=omega, quiet=1)
        ora = O.Oracle(W, H, D, solver=O.RBSOR, omega=omega, acc=acc)
    try:
        mask = np.zeros((D + 2, H + 2, W + 2), dtype=bool)
        mask[1:-1, 1:-1, 1:-1] = rng.random((D, H, W)) < rng.choice([0.0, 0.04, 0.25])
        sim.set_mask(mask)
        ora.set_mask(mask)
        for _ in range(2):
            sim.run_one()
            ora.run_one()
        for f in range(11):
            assert bits_equal(sim.get(f), ora.get(f)), (dims, acc, mode, omega, F.FIELD_NAMES[f])
    finally:
        sim.set_option("sweep_fuse", "3")
        sim.close()
