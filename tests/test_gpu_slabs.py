"""z-slab partitioning, checked on ONE GPU: 2 and 4 rank processes share the device and talk
through the host-staged shared-memory transport (RCCL refuses ranks on the same device).  They
run the product's slab code path -- wall flags, global-z flags and back-trace, halo exchange
after every sweep, all-gathered advection source, per-rank dump offsets, reduced statistics --
and must reproduce the single-GPU run bit for bit (halo exchange does not change arithmetic)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu
WORKER = os.path.join(ROOT, "tests", "slab_worker.py")


_REF_RUNS = {}          # single-GPU runs, shared by the cases that compare against the same one
_IPC = {}


def ipc_usable():
    """The FSIPC transport needs hipIpc memory handles between processes that share the GPU; where the pool refuses
    them the cases over it are skipped, with the probe's own words as the reason (tools/ipc_probe.cpp)."""
    if "ok" not in _IPC:
        exe = os.path.join(ROOT, "tools", "ipc_probe")
        if not os.path.exists(exe):
            _IPC["ok"], _IPC["why"] = False, "tools/ipc_probe was not built (python -c 'import __graft_entry__ as g; g.build()')"
        else:
            r = subprocess.run([exe, "2", "8", "1"], capture_output=True, text=True, timeout=120,
                               env=dict(os.environ, FS_IPC_TIMEOUT_S="20"))
            _IPC["ok"], _IPC["why"] = r.returncode == 0, (r.stdout + r.stderr)[-400:]
    return _IPC["ok"], _IPC["why"]


def run_ranks(tmp, nranks, args, env=None, transport="shm"):
    key = (tuple(str(a) for a in args), tuple(sorted((env or {}).items())))
    if nranks == 1 and key in _REF_RUNS:
        return _REF_RUNS[key]
    out = os.path.join(tmp, "n%d" % nranks)
    os.makedirs(os.path.join(out, "data"))
    idfile = os.path.join(out, "id.bin")
    if nranks > 1:
        import fluid_simulation_amd as F
        open(idfile, "wb").write(F.comm_unique_id(transport))
    procs = [subprocess.Popen([sys.executable, WORKER, str(r), str(nranks), idfile, out] + [str(a) for a in args],
                              env=dict(os.environ, FS_IPC_TIMEOUT_S="60", **(env or {}))) for r in range(nranks)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    if nranks == 1:
        _REF_RUNS[key] = out
    return out


SCHEDULES = [
    # (ranks, depth, options of the slab run, transport, solver iterations)
    (2, 16, "", "shm", 5), (4, 16, "", "shm", 5), (2, 32, "", "shm", 5), (3, 48, "", "shm", 5),
    (2, 32, "overlap=0", "shm", 5), (2, 32, "overlap=1", "shm", 7), (2, 32, "overlap=2", "shm", 7), (3, 48, "overlap=2", "shm", 8),
    # the same over the asynchronous device-to-device transport: nothing below host-synchronises, so a missing
    # stream dependency shows (7 and 8 iterations plan [3, 3, 1] and [3, 3, 2]: passes with more levels than the next)
    (2, 32, "", "ipc", 5), (2, 32, "overlap=0", "ipc", 7), (2, 32, "overlap=1", "ipc", 7), (2, 32, "overlap=2", "ipc", 7),
    (3, 48, "overlap=2", "ipc", 8), (4, 64, "", "ipc", 8), (3, 48, "comm_cus=auto", "ipc", 7), (2, 32, "comm_cus=8,overlap=1", "ipc", 5),
    (2, 32, "split_density_solve=0", "ipc", 5), (4, 16, "", "ipc", 5),
    # overlap=3, the push schedule: solver passes store their boundary planes straight into the neighbours' halo planes
    # (three-sweep, two-sweep and single-sweep kernels: 8 iterations plan [3, 3, 2], 7 plan [3, 3, 1]; thin slabs too)
    (2, 32, "overlap=3", "ipc", 8), (3, 48, "overlap=3", "ipc", 7), (4, 16, "overlap=3", "ipc", 5), (4, 64, "overlap=3", "ipc", 8),
    (2, 32, "overlap=3", "shm", 5),          # a transport that cannot push runs schedule 3 as schedule 0
]


@pytest.mark.parametrize("nranks,D,opts,transport,acc", SCHEDULES)
def test_slabs_match_single_gpu_bit_exact(tmp_path, nranks, D, opts, transport, acc):
    """Every communication schedule (overlap = 0 / 1 / 2 and "auto", with and without CUs kept free, with and without the
    split density solve) over both development transports: local depth 16 (D = 32, 48, 64) is deep enough for the
    boundary-first schedules, depth 8 and 4 take the plain path.  The options only reach the slab run: the single-GPU run
    it is compared with is the same for every schedule."""
    if transport == "ipc":
        ok, why = ipc_usable()
        if not ok:
            pytest.skip("FSIPC transport not usable on this box: " + why)
    W, H, steps = 20, 12, 3
    stl = os.path.join(GOLDEN, "sphere_24x12.stl")
    args = [W, H, D, acc, steps, stl, "fp32", "jacobi", opts]
    ref_dir = run_ranks(str(tmp_path), 1, [W, H, D, acc, steps, stl, "fp32", "jacobi", ""])
    par_dir = run_ranks(str(tmp_path), nranks, args, transport=transport)
    ref = np.load(os.path.join(ref_dir, "rank0.npz"))
    Dl = D // nranks
    for r in range(nranks):
        z = np.load(os.path.join(par_dir, "rank%d.npz" % r))
        zoff = int(z["zoff"])
        assert zoff == r * Dl
        for k in ("dens", "v_x", "v_y", "v_z", "obs", "pressure"):
            got, want = z[k], ref[k][zoff:zoff + Dl + 2]
            lo = 0 if r == 0 else 1               # interior planes + the physical ghost planes this rank owns
            hi = Dl + 2 if r == nranks - 1 else Dl + 1
            assert np.array_equal(got[lo:hi].view(np.uint32), want[lo:hi].view(np.uint32)), (r, k)
            if k != "pressure":
                # halo planes hold the neighbour's boundary planes after the last exchange
                assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (r, k, "halo")
        # the advection source was fetched through the velocity-bounded window (poisoned outside)
        assert 2 <= int(z["reach"]) < D
        # sum/min/max are all-reduced: every rank reports the global values (sum up to rounding order)
        assert np.allclose(z["stats"], ref["stats"], rtol=1e-12, atol=1e-12)
        # the slab steps never synchronised the compute stream: every reach came through the asynchronous path, two per
        # step (+ the one posted at the start of a step whose v_z_prev is not known from the step before) -- also across
        # the host-side edits that make v_z jump (setVelocity with v_z = 2.0 between steps)
        syncs, waits, blocked, plan, cus = (int(v) for v in z["sched"])
        assert syncs == 0 and waits == 2 * steps + 1 and 0 <= blocked <= waits, (r, z["sched"])
        assert int(z["sched_end"][0]) == 0, (r, z["sched_end"])
        want_plan = {"overlap=0": 0, "overlap=1": 1, "overlap=2": 2, "overlap=3": 3}
        forced = [v for k, v in want_plan.items() if k in opts]
        assert plan in (0, 1, 2, 3) and (not forced or plan == forced[0]), (r, plan)
        assert cus == (8 if "comm_cus=8" in opts else cus) and cus in (0, 8)
        plans = plans + [(plan, cus)] if r else [(plan, cus)]
    assert len(set(plans)) == 1, plans                      # "auto": every rank took the same decision
    # frame dumps: ranks wrote their planes at their own offsets of the same five files
    for fn in ("data", "obs", "v_x", "v_y", "v_z"):
        a = np.fromfile(os.path.join(ref_dir, "data", fn + ".bin"), dtype=np.uint8)
        b = np.fromfile(os.path.join(par_dir, "data", fn + ".bin"), dtype=np.uint8)
        assert a.size == steps * (W + 2) * (H + 2) * (D + 2) * 4
        assert np.array_equal(a, b), fn


@pytest.mark.parametrize("W,H,D,nranks,precision,solver,opts",
                         [(300, 9, 24, 2, "fp32", "jacobi", ""), (20, 12, 16, 2, "fp64", "jacobi", ""), (520, 7, 36, 3, "fp32", "jacobi", ""),
                          (24, 11, 32, 2, "fp32", "rbsor", ""), (20, 9, 12, 3, "fp64", "rbsor", ""),
                          (520, 7, 36, 3, "fp32", "jacobi", "two_sweep_kernel=fused"), (1000, 8, 24, 2, "fp32", "jacobi", "two_sweep_kernel=fused"),
                          (20, 12, 32, 2, "fp64", "jacobi", "two_sweep_kernel=fused"), (300, 9, 32, 2, "fp64", "jacobi", "two_sweep_kernel=fused"),
                          (20, 12, 16, 2, "fp32", "jacobi", "two_sweep_kernel=pair"), (512, 8, 32, 2, "fp32", "jacobi", ""),
                          # the push schedule (overlap=3, FSIPC) through every solver kernel: pair kernel on 1000-cell rows, fused
                          # two-sweep kernel, fp64, three sweeps on 512-cell rows, and rbsor (its passes fall back to copies)
                          (1000, 8, 24, 2, "fp32", "jacobi", "overlap=3,two_sweep_kernel=pair"), (520, 7, 36, 3, "fp32", "jacobi", "overlap=3,two_sweep_kernel=fused"),
                          (300, 9, 32, 2, "fp64", "jacobi", "overlap=3"), (512, 8, 32, 2, "fp32", "jacobi", "overlap=3"),
                          (24, 11, 32, 2, "fp32", "rbsor", "overlap=3"),
                          # lane-aligned rows on slabs: the three-sweep kernel with two bodies per group of iterations (an inner
                          # rank has no z wall: wall-free body throughout; the outer ranks one each), all schedules' kernels
                          (256, 30, 48, 3, "fp32", "jacobi", ""), (512, 20, 64, 4, "fp32", "jacobi", "overlap=3"), (256, 9, 32, 2, "fp32", "jacobi", "overlap=1,wall_free=0")])
def test_slabs_wide_rows_and_fp64(tmp_path, W, H, D, nranks, precision, solver, opts):
    """More than one 256-cell chunk per row, fp64 fields, each of the solver kernels on a slab (fp32 rows up to
    512 cells: three sweeps per pass across three-deep halos; the fused and the pair two-sweep kernel), and the
    optional red-black SOR solver (cell colour follows the global z, so slabs must agree with one GPU)."""
    args = [W, H, D, 7 if opts or W in (256, 512) else 4, 2, os.path.join(GOLDEN, "plate_ascii.stl"), precision, solver, opts]
    transport = "ipc" if ("overlap=3" in opts or W == 256) else "shm"
    if transport == "ipc":
        ok, why = ipc_usable()
        if not ok:
            pytest.skip("FSIPC transport not usable on this box: " + why)
    ref_dir = run_ranks(str(tmp_path), 1, args)
    par_dir = run_ranks(str(tmp_path), nranks, args, transport=transport)
    ref = np.load(os.path.join(ref_dir, "rank0.npz"))
    Dl = D // nranks
    u = np.uint64 if precision == "fp64" else np.uint32
    for r in range(nranks):
        z = np.load(os.path.join(par_dir, "rank%d.npz" % r))
        zoff = int(z["zoff"])
        for k in ("dens", "v_x", "v_y", "v_z", "obs"):
            got, want = z[k], ref[k][zoff:zoff + Dl + 2]
            assert got.dtype == want.dtype
            assert np.array_equal(got.view(u), want.view(u)), (r, k)
        triple_plan, fused2, zh = (int(v) for v in z["kernels"])
        if solver == "jacobi":
            assert zh == (3 if precision == "fp32" and W <= 512 else 2)
            assert (triple_plan >= 0) == (zh == 3)      # slab ranks run three sweeps per pass wherever the kernel exists
            if "fused" in opts:
                assert fused2 == 1


def test_depth_must_divide(tmp_path):
    import fluid_simulation_amd as F
    sim = F.Simulation(8, 8, 9, 1, quiet=1)
    with pytest.raises(F.FluidsimError):
        sim.comm_init(0, 2, F.comm_unique_id("shm"))


def _soak_cases():
    """Seeded random slab configurations: rank counts, slab depths down to the halo depth, row widths around the
    kernels' limits (256, 512, 1024 cells), both precisions, every communication schedule (incl. the push schedule and
    "auto"), both development transports, CU masks and two-sweep kernels.  FS_SOAK=N draws N
    cases instead of the default handful (development soak; every case is a full bit-exact comparison)."""
    n = int(os.environ.get("FS_SOAK", "6"))
    rng = np.random.default_rng(20261005)
    cases = []
    for i in range(n):
        nranks = int(rng.choice([2, 2, 3, 4]))
        dl = int(rng.choice([3, 4, 5, 7, 10, 14, 16, 19]))
        W = int(rng.choice([rng.integers(11, 48), rng.integers(250, 262), rng.integers(505, 520), rng.integers(1015, 1030)],
                           p=[0.5, 0.2, 0.2, 0.1]))
        H = int(rng.integers(7, 26))
        acc = int(rng.integers(1, 10))
        prec = str(rng.choice(["fp32", "fp32", "fp64"]))
        opts = ["overlap=%s" % rng.choice(["0", "1", "2", "3", "auto"]), "two_sweep_kernel=%s" % rng.choice(["auto", "pair", "fused"])]
        if rng.random() < 0.25:
            opts.append("sweep_fuse=2")
        if rng.random() < 0.2:
            opts.append("comm_cus=8")
        transport = str(rng.choice(["ipc", "ipc", "shm"]))
        cases.append((i, W, H, dl * nranks, nranks, acc, prec, ",".join(opts), transport))
    return cases


@pytest.mark.parametrize("case,W,H,D,nranks,acc,precision,opts,transport", _soak_cases())
def test_random_slab_configurations_match_single_gpu(tmp_path, case, W, H, D, nranks, acc, precision, opts, transport):
    if transport == "ipc":
        ok, why = ipc_usable()
        if not ok:
            pytest.skip("FSIPC transport not usable on this box: " + why)
    args = [W, H, D, acc, 2, os.path.join(GOLDEN, "plate_ascii.stl"), precision, "jacobi", opts]
    ref_dir = run_ranks(str(tmp_path), 1, args)
    par_dir = run_ranks(str(tmp_path), nranks, args, transport=transport)
    ref = np.load(os.path.join(ref_dir, "rank0.npz"))
    Dl = D // nranks
    u = np.uint64 if precision == "fp64" else np.uint32
    for r in range(nranks):
        z = np.load(os.path.join(par_dir, "rank%d.npz" % r))
        zoff = int(z["zoff"])
        for k in ("dens", "v_x", "v_y", "v_z", "obs", "pressure"):
            got, want = z[k], ref[k][zoff:zoff + Dl + 2]
            lo = 0 if r == 0 else 1
            hi = Dl + 2 if r == nranks - 1 else Dl + 1
            assert np.array_equal(got[lo:hi].view(u), want[lo:hi].view(u)), (case, r, k)


def test_config4_grid_split_into_slabs_matches_single_gpu(tmp_path):
    """BASELINE config 4's grid at its full size -- 1024x512x512, rows of 1024 cells, 2-deep halos, two sweeps per pass --
    split into 4 z-slabs of 128 planes (ranks share the one GPU, halo planes through host shared memory), against the
    same run whole on the GPU: every plane of every field, compared through per-plane SHA-256 digests (the arrays are
    6.5 GB per run).  What remains untested of config 4 is the RCCL transport itself between 8 devices."""
    W, H, D, nranks = 1024, 512, 512, 4
    args = [W, H, D, 3, 1, os.path.join(GOLDEN, "plate_ascii.stl"), "fp32", "jacobi", "dump_every=0"]
    env = {"FS_SLAB_DIGEST": "1"}
    ref_dir = run_ranks(str(tmp_path), 1, args, env)
    par_dir = run_ranks(str(tmp_path), nranks, args, env)
    ref = np.load(os.path.join(ref_dir, "rank0.npz"))
    Dl = D // nranks
    for r in range(nranks):
        z = np.load(os.path.join(par_dir, "rank%d.npz" % r))
        zoff = int(z["zoff"])
        assert zoff == r * Dl and int(z["kernels"][2]) == 2
        for k in ("dens", "v_x", "v_y", "v_z", "obs", "pressure"):
            got, want = z[k], ref[k][zoff:zoff + Dl + 2]
            lo = 0 if r == 0 else 1
            hi = Dl + 2 if r == nranks - 1 else Dl + 1
            assert got.shape == (Dl + 2, 32)
            assert np.array_equal(got[lo:hi], want[lo:hi]), (r, k)
        assert np.allclose(z["stats"], ref["stats"], rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("W,H,D,nranks,precision,opts,transport",
                         [(32, 16, 32, 2, "fp32", "mg_min_planes=4", "shm"),      # both coarse levels distributed (8 and 4 planes per rank)
                          (32, 16, 32, 2, "fp32", "mg_min_planes=8", "ipc"),      # level 1 distributed, level 2 held whole by every rank
                          (32, 16, 32, 2, "fp32", "mg_min_planes=16", "ipc"),     # every coarse level held whole (the seam is level 0 -> 1)
                          (32, 16, 64, 4, "fp32", "", "ipc"), (32, 16, 64, 4, "fp64", "mg_min_planes=8,overlap=3", "ipc"),
                          (64, 32, 48, 3, "fp32", "mg_min_planes=2", "shm"), (24, 16, 32, 2, "fp32", "mg_cycles=2,mg_pre=2,mg_post=2,overlap=1", "ipc")])
def test_slabs_multigrid_matches_single_gpu(tmp_path, W, H, D, nranks, precision, opts, transport):
    """solver=mg on z-slabs (round 3): the pressure equation's V-cycles with the coarse levels distributed like level 0 (one
    halo exchange per red-black colour) down to `mg_min_planes` planes per rank and held whole by every rank below that
    (all-gather at the seam) -- same operations in the same order as on one GPU, so every field must come out bit-identical,
    also across the obstacle edits between steps that rebuild the coarse operators."""
    if transport == "ipc":
        ok, why = ipc_usable()
        if not ok:
            pytest.skip("FSIPC transport not usable on this box: " + why)
    stl = os.path.join(GOLDEN, "plate_ascii.stl")
    ref_dir = run_ranks(str(tmp_path), 1, [W, H, D, 5, 2, stl, precision, "mg", ",".join(o for o in opts.split(",") if o.startswith("mg_c") or o.startswith("mg_p"))])
    par_dir = run_ranks(str(tmp_path), nranks, [W, H, D, 5, 2, stl, precision, "mg", opts], transport=transport)
    ref = np.load(os.path.join(ref_dir, "rank0.npz"))
    Dl = D // nranks
    u = np.uint64 if precision == "fp64" else np.uint32
    for r in range(nranks):
        z = np.load(os.path.join(par_dir, "rank%d.npz" % r))
        zoff = int(z["zoff"])
        for k in ("dens", "v_x", "v_y", "v_z", "obs", "pressure"):
            got, want = z[k], ref[k][zoff:zoff + Dl + 2]
            lo = 0 if r == 0 else 1
            hi = Dl + 2 if r == nranks - 1 else Dl + 1
            assert np.array_equal(got[lo:hi].view(u), want[lo:hi].view(u)), (r, k)
        assert int(z["sched"][0]) == 0
    assert np.abs(ref["pressure"]).max() > 0


def test_bench_slab_parity_rehearsal_repeated_handles(tmp_path):
    """bench.py's own N > 1 parity check (every communication schedule in turn, each on a fresh slab handle and a fresh
    FSIPC transport), four rank processes sharing the GPU, twice over -- ten slab handles per process, launched exactly
    like the bench (torch.distributed.run, torch imported first).  Round 3 found this way that many small exported
    allocations alias after a few handles of one process (wrong planes, then hipIpcGetMemHandle: invalid argument); the
    library now exports one arena per handle and verifies every mapping."""
    ok, why = ipc_usable()
    if not ok:
        pytest.skip("FSIPC transport not usable on this box: " + why)
    port = 29700 + os.getpid() % 200
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "tools", "parity_rehearsal.py"), "ipc", "2"],
                       capture_output=True, text=True, timeout=600, cwd=str(tmp_path),
                       env=dict(os.environ, FS_IPC_TIMEOUT_S="60", HSA_ENABLE_IPC_MODE_LEGACY="0"))
    lines = [l for l in r.stdout.splitlines() if l.startswith("repeat")]
    assert r.returncode == 0 and len(lines) == 2, (r.stdout[-1500:], r.stderr[-1500:])
    for l in lines:
        assert " True " in l and "'3': 3" in l, l
