#!/usr/bin/env python3
"""bench.py -- the north-star metric on synthetic wind tunnels (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload auto|c2|c3|c4]

A "step" is one iteration of the reference's time loop (Simulation::run(), simulation.cpp:63-71:
inlet density, buffer = dens, Simulation::step()) with frame dumps off.  Fields live in HBM for
the whole timed region.  Workloads (BASELINE.json configs):
    c2  256^3,          sphere,         40 solver iterations
    c3  512^3,          sphere + plate, 80 solver iterations   <- N=1 default ("roofline run")
    c4  1024x512x512,   sphere + plate, 80 solver iterations   <- N>1 default, z-slabs, strong scaling
For N>1 the driver launches one rank per GPU with torch.distributed.run; ranks exchange halo
planes over RCCL inside libfluidsim.so.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
SWEEP_BYTES_PER_CELL = 12    # read x, read rhs, write x (fp32)

WORKLOADS = {
    "c2": dict(W=256, H=256, D=256, acc=40, plate=False),
    "c3": dict(W=512, H=512, D=512, acc=80, plate=True),
    "c4": dict(W=1024, H=512, D=512, acc=80, plate=True),
}


def add_obstacles(F, sim, cfg, tmp):
    """Synthetic STL meshes through the reference's loader path (object_loader.cpp:270-452).
    Mesh units are chosen so the loader samples at its finest grid (200^3)."""
    from fluid_simulation_amd import shapes
    W, H, D = cfg["W"], cfg["H"], cfg["D"]
    sphere = shapes.write_binary_stl(os.path.join(tmp, "sphere.stl"), shapes.sphere_triangles(2.0, 48, 24))
    added = [F.loadSTLIntoObstacles(sphere, sim, 0.3, 0.0, 0.0, 0.0, -W / 4.0, 0.0, 0.0)]
    if cfg["plate"]:
        plate = shapes.write_binary_stl(os.path.join(tmp, "plate.stl"), shapes.box_triangles(0.2, 2.4, 1.6))
        added.append(F.loadSTLIntoObstacles(plate, sim, 0.45, 0.0, 0.0, 0.0, W / 8.0, 0.0, 0.0))
    return added


def kernels_sha():
    """Identity of the kernel sources a PMC measurement belongs to (profiles/sweep_traffic.json stamps)."""
    import hashlib
    h = hashlib.sha256()
    for fn in ("kernels.hip", "sweep_fused.hip", "kernels.h", "kernels_dev.h"):
        with open(os.path.join(ROOT, "fluid_simulation_amd", "csrc", fn), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def pressure_residual(p, div, obs):
    """|| div + sum of the six neighbours - 6 p || over fluid cells, relative to || div || (the equation
    linearSolver(0, p, div, 1, 6) iterates on, simulation.cpp:320): how far a fixed iteration count got."""
    import numpy as np
    p = p.astype(np.float64)
    c = p[1:-1, 1:-1, 1:-1]
    nb = (p[1:-1, 1:-1, 2:] + p[1:-1, 1:-1, :-2] + p[1:-1, 2:, 1:-1] + p[1:-1, :-2, 1:-1] + p[2:, 1:-1, 1:-1] + p[:-2, 1:-1, 1:-1])
    r = div[1:-1, 1:-1, 1:-1].astype(np.float64) + nb - 6.0 * c
    fluid = obs[1:-1, 1:-1, 1:-1] < 0.5
    d = div[1:-1, 1:-1, 1:-1][fluid].astype(np.float64)
    return float(np.sqrt((r[fluid] ** 2).sum()) / max(1e-300, np.sqrt((d ** 2).sum())))


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(budget_s=10.0):
    """The reference CPU path timed on this box's host cores, on the two quoted configurations a CPU can
    finish (the GPU workload itself, 512^3 / 80 iterations, is ~10 minutes per step on 16 cores):
      * BASELINE config 1 exactly -- 64^3 empty tunnel, 50 steps, 20 iterations, constructor defaults --
        on all granted cores, and (bounded to `budget_s`) on one thread, the reference's only deterministic
        configuration;
      * BASELINE config 2's grid -- 256^3, ball obstacle, 40 iterations -- ONE step on all granted cores:
        the sample `value` is quoted on, next to which the line's `extra_256` is the GPU on the same config.
    Uses the compiled reference (oracle/_ref/libref.so, kind "reference") when it travelled with the
    repo, else the C restatement (kind "port")."""
    import numpy as np
    from oracle import cpu_ref as O
    O.build()
    # the box's CPU share for one GPU is 16 cores even where os.cpu_count() reports the whole host
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, int(os.environ.get("FS_CPU_BASELINE_THREADS", "16"))))
    kind = "reference" if O.have_reference() else "port"

    def make(W, H, D, threads, acc):
        if kind == "reference":
            return O.Reference(W, H, D, threads=threads, iter=1, acc=acc)
        return O.Oracle(W, H, D, solver=O.GS_LEX, threads=threads, iter=1, acc=acc)

    def run_steps(sim, nmax, budget):
        n, t0 = 0, time.perf_counter()
        while n < nmax:
            sim.run_one()
            n += 1
            if time.perf_counter() - t0 >= budget:
                break
        return n, time.perf_counter() - t0

    # config 1, all cores: the full 50 steps (a few seconds)
    sim = make(64, 64, 64, cores, 20)
    n_all, el_all = run_steps(sim, 50, 120.0)
    sim.close()
    # config 1, one thread: as many of the 50 steps as fit the budget
    sim = make(64, 64, 64, 1, 20)
    n_one, el_one = run_steps(sim, 50, budget_s)
    sim.close()
    c1 = 64 ** 3
    out = {
        "config1": {"workload": "64x64x64 empty tunnel, 20 iterations (BASELINE config 1)",
                    "cells_steps_per_sec": c1 * n_all / el_all, "steps": n_all, "seconds": el_all, "threads": cores,
                    "cells_steps_per_sec_1_thread": c1 * n_one / el_one, "steps_1_thread": n_one, "seconds_1_thread": el_one},
        "cores": cores, "kind": kind, "unit": "cells*steps/s",
        "cpu_model": cpu_model(), "host_cpus_visible": os.cpu_count(),
    }
    if budget_s >= 5.0:
        # config 2's grid, one step (10-30 s of CPU work on 16 cores)
        W = H = D = 256
        z, y, x = np.mgrid[0:D + 2, 0:H + 2, 0:W + 2]
        mask = ((x - 64) ** 2 + (y - 128) ** 2 + (z - 128) ** 2) <= 38 ** 2
        del z, y, x
        sim = make(W, H, D, cores, 40)
        sim.set_mask(mask)
        del mask
        n2, el2 = run_steps(sim, 1, 0.0)
        sim.close()
        out["value"] = W * H * D * n2 / el2
        out["sample"] = ("256x256x256 tunnel, analytic ball obstacle r=38 (a numpy mask, not the STL sphere), 40 iterations "
                         "(BASELINE config 2's grid), 1 step in "
                         "%.1f s, OpenMP %d threads, dumps off; config1 = 64x64x64 empty, 20 iterations, %d steps in %.1f s"
                         % (el2, cores, n_all, el_all))
    else:                                                # short form (tests): config 1 only
        out["value"] = out["config1"]["cells_steps_per_sec"]
        out["sample"] = "64x64x64 empty tunnel, 20 iterations (BASELINE config 1), %d steps in %.1f s, OpenMP %d threads, dumps off" % (
            n_all, el_all, cores)
    out["value_1_thread"] = out["config1"]["cells_steps_per_sec_1_thread"]
    return out


def slab_parity_check(F, fsdist, dist, rank, world, transport, ctl_device, schedules=None):
    """N>1 only, before the timed run: every rank runs a small tunnel whole on its own GPU, then as its z-slab of
    a world-wide run over the real transport -- once per communication schedule (overlap = auto, 0, 1, 2: "auto" times
    the other three and every rank must come out with the same one) -- and compares its planes of every field bit for
    bit.  This is the multi-rank parity test over RCCL that a one-GPU box cannot run (tests/test_gpu_slabs.py does the
    same over the two development transports)."""
    import numpy as np
    W, H, acc, steps = 96, 40, 7, 3
    D = 16 * world                                       # 16 planes per rank: deep enough for the boundary-first overlapped exchange
    z, y, x = np.mgrid[0:D + 2, 0:H + 2, 0:W + 2]
    mask = ((x - W / 3.0) ** 2 + (y - H / 2.0) ** 2 + (z - (D / 2.0 + 1.5)) ** 2) <= (min(H, D) / 3.0) ** 2
    mask[0] = mask[-1] = False
    mask[:, 0] = mask[:, -1] = False
    mask[:, :, 0] = mask[:, :, -1] = False
    fields = (F.DENS, F.VX, F.VY, F.VZ, F.PRESSURE, F.DIVERGENCE)
    whole = F.Simulation(W, H, D, steps, acc=acc, quiet=1, dump_every=0)
    whole.set_mask(mask)
    for _ in range(steps):
        whole.run_one()
    want = {f: whole.get(f) for f in fields}
    ws = whole.stats(F.VX)                               # (sum, min, max) over the global grid
    whole.close()
    bad, ran = [], {}
    for overlap in (schedules or ("auto", "0", "1", "2") + (("3",) if transport == "ipc" else ())):
        slab = F.Simulation(W, H, D, steps, acc=acc, quiet=1, dump_every=0, overlap=overlap)
        uid = fsdist.share_unique_id(dist, lambda: F.comm_unique_id(transport), rank, device=ctl_device)
        slab.comm_init(rank, world, uid)
        dl, zoff = slab.local_depth, slab.z_offset
        slab.set_mask(mask[zoff:zoff + dl + 2])
        for _ in range(steps):
            slab.run_one()
        for f in fields:
            if want[f][zoff + 1:zoff + dl + 1].tobytes() != slab.get(f)[1:dl + 1].tobytes():
                bad.append("%s (overlap=%s)" % (F.FIELD_NAMES[f], overlap))
        ss = slab.stats(F.VX)
        if not (ss[1:] == ws[1:] and abs(ss[0] - ws[0]) <= 1e-9 * max(1.0, abs(ws[0]))):
            bad.append("stats (overlap=%s)" % overlap)
        plan = slab._geti("overlap_plan")
        if fsdist.max_over_ranks(dist, float(plan), device=ctl_device) != -fsdist.max_over_ranks(dist, -float(plan), device=ctl_device):
            bad.append("ranks disagree on the schedule (overlap=%s)" % overlap)
        if slab._geti("stream_syncs") != 0:
            bad.append("a slab step synchronised the compute stream (overlap=%s)" % overlap)
        ran[overlap] = plan
        slab.close()
    flag = fsdist.max_over_ranks(dist, 1.0 if bad else 0.0, device=ctl_device)
    if bad:
        sys.stderr.write("rank %d: z-slab run differs from the single-GPU run: %s\n" % (rank, bad))
    return {"ok": flag == 0.0, "grid": [W, H, D], "steps": steps, "acc": acc, "schedule_run_under_overlap": ran,
            "what": "every rank: its planes of dens/v/p/div of a %d-rank slab run vs the same run whole on its own GPU, bit-exact, "
                    "once per communication schedule (%s)" % (world, ", ".join(ran))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="auto", choices=["auto", "c2", "c3", "c4"])
    ap.add_argument("--precision", default="fp32", choices=["fp32", "fp64"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=10.0,
                    help="seconds for the one-thread leg of cpu_baseline; below 5 the 256^3 sample is skipped (tests)")
    ap.add_argument("--launch-plans", default=None,
                    help="\"<two-sweep plan id>,<three-sweep plan id>\": replay the launch plans of another run instead of timing "
                         "them (tools/make_profiles.sh: counter passes must run what the bench line ran)")
    ap.add_argument("--transport", default="rccl", choices=["rccl", "ipc", "shm"],
                    help="rccl (default): halo planes over RCCL/xGMI, one GPU per rank.  ipc: stream-ordered device-to-device "
                         "copies between the rank processes (csrc/ipc.h); with one GPU per rank an alternative to RCCL, with "
                         "fewer GPUs than ranks a development rehearsal (ranks share GPUs; never a result).  shm: the host-"
                         "staged synchronous development transport; never a result")
    ap.add_argument("--overlap", default="auto", choices=["auto", "0", "1", "2", "3"],
                    help="communication schedule of the slab passes (fs_set_option \"overlap\"); auto = timed over the real transport")
    ap.add_argument("--comm-cus", default="auto",
                    help="CUs kept free of solver workgroups for the transport's kernels: auto (default: none and 8 are both timed with "
                         "the schedules, the slowest rank's time decides), 0, or N")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X (no HIP device visible); the solver has no CPU path")
    if args.transport == "rccl" and torch.cuda.device_count() < world:
        sys.exit("bench.py --gpus %d over RCCL needs %d GPUs (one per rank; %d visible): RCCL refuses ranks that share a device.  "
                 "--transport ipc rehearses the N > 1 path with ranks sharing GPUs (never a result)" % (world, world, torch.cuda.device_count()))
    rehearsal = args.transport == "shm" or (args.transport == "ipc" and torch.cuda.device_count() < world)
    if rehearsal:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    ctl_device = None if rehearsal else torch.device("cuda", local_rank)   # where control-plane tensors live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import fluid_simulation_amd as F

    name = args.workload
    if name == "auto":
        name = "c3" if world == 1 else "c4"
    cfg = WORKLOADS[name]
    W, H, D, acc = cfg["W"], cfg["H"], cfg["D"], cfg["acc"]
    cells = W * H * D

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def trace(msg):
        if os.environ.get("FS_BENCH_TRACE"):
            sys.stderr.write("bench rank %d: %s\n" % (rank, msg))
            sys.stderr.flush()

    slab_parity = None
    if world > 1:
        from fluid_simulation_amd import dist as fsdist
        if os.environ.get("FS_BENCH_SKIP_PARITY"):       # development only: the line then carries slab_parity = null
            trace("parity check SKIPPED (FS_BENCH_SKIP_PARITY)")
        else:
            slab_parity = slab_parity_check(F, fsdist, dist, rank, world, args.transport, ctl_device)

    trace("parity check done: %s" % (slab_parity,))
    sim = F.Simulation(W, H, D, args.steps, acc=acc, precision=args.precision, quiet=1, dump_every=0, profile=1,
                       overlap=args.overlap, comm_cus=args.comm_cus)
    if args.launch_plans:
        sim.set_option("launch_plans", args.launch_plans)
    if world > 1:
        uid = fsdist.share_unique_id(dist, lambda: F.comm_unique_id(args.transport), rank, device=ctl_device)
        sim.comm_init(rank, world, uid)
    with tempfile.TemporaryDirectory() as tmp:
        added = add_obstacles(F, sim, cfg, tmp)

    trace("obstacles in: %s" % (added,))
    for _ in range(args.warmup):
        sim.run_one()
        trace("warm-up step queued")
    sim.sync()
    trace("warm-up done")
    sim.reset_timing()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sim.run_one()
    sim.sync()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        elapsed = fsdist.max_over_ranks(dist, elapsed, device=ctl_device)

    # dominant kernel: the solver sweep, HIP events on the solver's own stream over the timed
    # region (csrc/sweep_fused.hip: three iterations per launch for fp32 rows up to 512 cells, two for
    # fp64 and for rows of 1024 cells; temporal blocking).
    fams = ("sweep", "sweep_pair", "sweep_triple", "divergence", "gradient", "advect", "misc", "comm")
    fam = {k: sim.timing(k) for k in fams}
    local_cells = W * H * sim.local_depth
    pair_shape = sim._geti("pair_shape")
    elem = 8 if args.precision == "fp64" else 4
    pair_ms, pair_n = fam["sweep_pair"]
    one_ms, one_n = fam["sweep"]
    tri_ms, tri_n = fam["sweep_triple"]
    # the dominant kernel is whichever solver kernel the time went to: three sweeps per launch where the
    # host driver found that faster on this grid, else two (z-slab ranks: always two), else one
    two_name = "jacobi_fused_kernel<NL=2>" if sim._geti("two_sweep_fused") else "jacobi_pair_kernel"
    if tri_ms >= pair_ms and tri_n > 0:
        kernel, iters_per_launch, k_ms, k_n = "jacobi_fused_kernel<NL=3>", 3, tri_ms, tri_n
    elif pair_n > 0:
        kernel, iters_per_launch, k_ms, k_n = two_name, 2, pair_ms, pair_n
    else:
        kernel, iters_per_launch, k_ms, k_n = "jacobi_sweep_kernel", 1, one_ms, one_n
    bytes_per_launch = SWEEP_BYTES_PER_CELL * (elem // 4) * local_cells * iters_per_launch
    avg_ms = k_ms / max(1, k_n)
    achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if k_n else 0.0
    total_iters = 3 * tri_n + 2 * pair_n + one_n
    iters_per_sec = total_iters / ((tri_ms + pair_ms + one_ms) * 1e-3) if total_iters else None

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    # roofline.traffic: HBM bytes per launch of the dominant kernel from the committed PMC passes -- only when
    # they were taken on THIS code and THIS launch plan (profiles/sweep_traffic.json carries a stamp: hash of
    # the kernel sources, grid, workgroup shape and z-chunk plan); anything else prints null, never a stale number
    traffic, traffic_note = None, "no PMC measurement committed for this workload/kernel"
    triple_plan = sim._geti("triple_plan")
    tpath = os.path.join(ROOT, "profiles", "sweep_traffic.json")
    tkey = name + ("_fp64" if args.precision == "fp64" else "")
    if os.path.exists(tpath):
        try:
            with open(tpath) as f:
                entry = json.load(f).get(tkey, {}).get(kernel, {})
            stamp = entry.get("stamp") or {}
            want = {"kernels_sha": kernels_sha(), "grid": [W, H, D], "pair_shape": pair_shape, "triple_plan": triple_plan}
            if entry and all(stamp.get(k) == v for k, v in want.items()):
                traffic = entry.get("hbm_bytes_per_launch")
                traffic_note = "rocprofv3 PMC (2*FETCH_SIZE + WRITE_SIZE, KiB units), measured at commit %s" % stamp.get("commit", "?")
            elif entry:
                traffic_note = "committed PMC measurement is for another build or launch plan (stamp %s, this run %s)" % (
                    json.dumps(stamp, sort_keys=True), json.dumps(want, sort_keys=True))
        except Exception as e:  # noqa: BLE001
            traffic_note = "could not read profiles/sweep_traffic.json: %s" % e

    out = {
        "metric": "cells_steps_per_sec",
        "value": cells * args.steps / elapsed,
        "unit": "cells*steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong" if world > 1 else None,   # one GPU: nothing scales (the N > 1 lines run config 4 strong-scaled)
        "vs_baseline": None,
        "dtype": "f64" if args.precision == "fp64" else "f32",
        "data": "synthetic",
        "config": {
            "workload": "%s: %dx%dx%d wind tunnel, %s, %d solver iterations per solve, Jacobi, dumps off"
                        % (name, W, H, D, "sphere + plate STL obstacles" if cfg["plate"] else "sphere STL obstacle", acc),
            "grid": [W, H, D], "acc": acc, "solver": "jacobi",
            "parallelism": ("z-slabs x%d, %s" % (world, "REHEARSAL, ranks share GPUs (not a result)" if rehearsal
                                                  else "RCCL halo exchange over xGMI" if args.transport == "rccl"
                                                  else "device-to-device copies between rank processes (FSIPC)")) if world > 1 else "single GPU",
            "voxelizer_points_added": added,
            "halo_transport": sim.comm_transport(),
            # N > 1 lines run config 4 strong-scaled; the N = 1 line runs config 3 (the roofline run) and carries the matching
            # single-GPU number of config 4 as extra_c4_single_gpu: divide by THAT for a scaling factor
            "strong_scaling_base": ("extra_c4_single_gpu.cells_steps_per_sec of the --gpus 1 line (config 4 on one GPU)"
                                    if world > 1 else None),
        },
        "jacobi_iter_per_sec": iters_per_sec,
        "roofline": {
            "kernel": kernel, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            # the fraction that cannot exceed 1: measured HBM bytes per launch / launch time / peak (null without a PMC
            # measurement of THIS build and launch plan); min_traffic_bytes = one read of iterate and right-hand side and
            # one write of the result per launch, whatever the number of iterations applied on the way
            "frac_physical": (traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (traffic and k_n) else None,
            "min_traffic_bytes": SWEEP_BYTES_PER_CELL * (elem // 4) * local_cells,
            "bytes_per_launch": bytes_per_launch, "solver_iterations_per_launch": iters_per_launch,
            "avg_launch_ms": avg_ms, "launches": k_n, "workgroup_shape_id": pair_shape,
            "launch_plan_three_sweeps": triple_plan, "two_sweep_kernel": two_name, "traffic_note": traffic_note,
            "note": "achieved = 12 B x cells x iterations per launch / HIP-event launch time; above the "
                    "physical HBM rate when several iterations share one pass over memory (temporal blocking); "
                    "traffic = measured HBM bytes per launch (rocprofv3 PMC, profiles/)",
        },
        "kernel_ms": {k: {"total_ms": v[0], "launches": v[1]} for k, v in fam.items()},
        "slab_parity": slab_parity,
        # N > 1: the communication schedule the run used (overlap "auto" = timed over the real transport, the slowest
        # rank's time per pass of each candidate decides) and what the slab steps cost the host
        "comm": None if world == 1 else {
            "overlap_plan": sim._geti("overlap_plan"), "comm_cus": sim._geti("comm_cus_plan"),
            "candidates_ms_per_pass": {"overlap=%d%s" % (k % 4, ", CU mask" if k >= 4 else ""): sim._getf("overlap%d_ms" % k)
                                       for k in range(8) if sim._getf("overlap%d_ms" % k) > 0},
            "comm_family_ms_per_step": fam["comm"][0] / args.steps,
            "comm_family_note": "halo refreshes outside the solver, advection gathers and reductions, waits included; the "
                                "solver's own exchanges fall inside the sweep families",
            "stream_syncs": sim._geti("stream_syncs"), "reach_waits": sim._geti("reach_waits"),
            "reach_waits_blocked": sim._geti("reach_waits_blocked"), "reach_wait_us": sim._geti("reach_wait_us"),
            "reach_hidden": sim._geti("reach_hidden"), "reach_exposed": sim._geti("reach_exposed"),
            "reach_note": "the reach of each advection gather arrives asynchronously (no stream synchronisation); hidden / exposed = "
                          "advections queued while the device was still busy with the half density solve placed before them / after it ran dry",
        },
        "step_bytes_per_cell_algorithmic": 208 + 72 * acc,
        "step_roofline_frac": (208 + 72 * acc) * (elem // 4) * cells * args.steps / elapsed / 1e9 / HBM_PEAK_GBS,
    }
    if world == 1 and not args.no_extra:
        # the pressure solve in isolation -- `acc` sweeps (a = 1, c = 6) on the divergence
        # field of the state the run ended in, repeated 20 times, HIP events around each repetition
        reps = [sim.time_sweeps(0, F.PRESSURE, F.DIVERGENCE, 1.0, 6.0, acc) for _ in range(20)]
        best, mean = min(reps), sum(reps) / len(reps)
        out["pressure_sweep_microbench"] = {
            "sweeps_per_repetition": acc, "repetitions": len(reps),
            "ms_per_sweep_mean": mean, "ms_per_sweep_best": best,
            "jacobi_iter_per_sec": 1e3 / mean,
            "GBps_algorithmic": SWEEP_BYTES_PER_CELL * (elem // 4) * cells / (mean * 1e-3) / 1e9,
        }
    sim.close()

    if world == 1 and not args.no_extra and name == "c3":
        # the other size the metric is quoted on: 256^3 (config 2), short run
        c2 = WORKLOADS["c2"]
        s2 = F.Simulation(c2["W"], c2["H"], c2["D"], 3, acc=c2["acc"], quiet=1, dump_every=0, profile=1)
        with tempfile.TemporaryDirectory() as tmp:
            add_obstacles(F, s2, c2, tmp)
        s2.run_one()
        s2.sync()
        s2.reset_timing()
        t0 = time.perf_counter()
        for _ in range(5):
            s2.run_one()
        s2.sync()
        e2 = time.perf_counter() - t0
        (p_ms, p_n), (o_ms, o_n), (t_ms, t_n) = s2.timing("sweep_pair"), s2.timing("sweep"), s2.timing("sweep_triple")
        it2 = (3 * t_n + 2 * p_n + o_n) / ((t_ms + p_ms + o_ms) * 1e-3)
        out["extra_256"] = {
            "workload": "c2: 256^3, sphere, 40 iterations",
            "cells_steps_per_sec": 256 ** 3 * 5 / e2,
            "jacobi_iter_per_sec": it2,
            "sweep_GBps_algorithmic": 12 * 256 ** 3 * it2 / 1e9,
        }
        # Equal-quality view of the headline (the solver here is Jacobi, the reference sweeps in place, i.e.
        # Gauss-Seidel at one thread): residual of the pressure equation of this developed flow after the same
        # number of iterations in both orders, and after twice as many Jacobi iterations.
        import numpy as np
        div2, obs2 = s2.get(F.DIVERGENCE), s2.get(F.OBS)
        q, q_ms = {}, {}
        for label, solver, n in (("jacobi_%d" % c2["acc"], "jacobi", c2["acc"]), ("jacobi_%d" % (2 * c2["acc"]), "jacobi", 2 * c2["acc"]),
                                 ("reference_order_%d" % c2["acc"], "gs_lex", c2["acc"]), ("multigrid_2_cycles", "mg", 2),
                                 ("multigrid_4_cycles", "mg", 4)):
            s2.set_option("solver", solver)
            if solver == "mg":
                s2.set_option("mg_cycles", n)
            else:
                s2.acc = n
            for rep in range(2):                         # the second one is timed (the first builds whatever the mode needs)
                s2.set(F.PRESSURE, np.zeros_like(div2))
                s2.set(F.DIVERGENCE, div2)
                s2.sync()
                t0 = time.perf_counter()
                s2.linear_solver(0, F.PRESSURE, F.DIVERGENCE, 1.0, 6.0)
                s2.sync()
                q_ms[label] = (time.perf_counter() - t0) * 1e3
            q[label] = pressure_residual(s2.get(F.PRESSURE), div2, obs2)
        out["solver_quality_256"] = {
            "relative_residual_after_iterations": q,
            "solve_ms": q_ms,
            "note": "pressure equation of the c2 flow after 6 steps, zero initial guess; reference_order = the reference's in-place "
                    "sweep at one thread (solver=gs_lex); the headline's Jacobi needs about twice the iterations for the same residual; "
                    "multigrid = the optional solver=mg (V-cycles; not the reference's arithmetic, not the headline)",
        }
        del div2, obs2
        s2.close()

    if world == 1 and not args.no_extra and name == "c3":
        # transparency: the reference's density diffusion is dead work (its result is overwritten by
        # the following advect, simulation.cpp:135-136).  The headline above executes it; this is
        # the same workload with it elided (identical fields, tests/test_gpu_parity.py).
        s3 = F.Simulation(W, H, D, 3, acc=acc, quiet=1, dump_every=0, elide_dead_density_solve=1)
        with tempfile.TemporaryDirectory() as tmp:
            add_obstacles(F, s3, cfg, tmp)
        s3.run_one()
        s3.sync()
        t0 = time.perf_counter()
        for _ in range(3):
            s3.run_one()
        s3.sync()
        out["extra_dead_density_solve_elided"] = {"cells_steps_per_sec": cells * 3 / (time.perf_counter() - t0),
                                                  "note": "not the headline; 5*acc instead of 6*acc sweeps per step"}
        s3.close()

    if world == 1 and not args.no_extra and name == "c3":
        # also with the last frame dumped (fs_run, dump_every = -1), writer flushed
        dump_root = "/dev/shm" if os.path.isdir("/dev/shm") else None
        with tempfile.TemporaryDirectory(dir=dump_root) as dtmp:
            s4 = F.Simulation(W, H, D, 3, acc=acc, quiet=1, dump_every=-1, dump_dir=dtmp)
            with tempfile.TemporaryDirectory() as tmp:
                add_obstacles(F, s4, cfg, tmp)
            s4.run_one()
            s4.sync()
            t0 = time.perf_counter()
            s4.run()
            s4.sync()
            out["extra_last_frame_dump"] = {"cells_steps_per_sec": cells * 3 / (time.perf_counter() - t0),
                                            "note": "3 steps through fs_run + one 2.7 GB frame written to " + (dump_root or "tmp")}
            s4.close()

    if world == 1 and not args.no_extra and name == "c3":
        # the N > 1 lines run config 4 (1024x512x512, strong scaling): the same workload on this one
        # GPU, so that whoever divides the multi-GPU values has the matching single-GPU number
        c4 = WORKLOADS["c4"]
        s5 = F.Simulation(c4["W"], c4["H"], c4["D"], 3, acc=c4["acc"], quiet=1, dump_every=0)
        with tempfile.TemporaryDirectory() as tmp:
            add_obstacles(F, s5, c4, tmp)
        s5.run_one()
        s5.sync()
        t0 = time.perf_counter()
        for _ in range(3):
            s5.run_one()
        s5.sync()
        out["extra_c4_single_gpu"] = {"workload": "c4: 1024x512x512, sphere + plate, 80 iterations",
                                      "cells_steps_per_sec": c4["W"] * c4["H"] * c4["D"] * 3 / (time.perf_counter() - t0),
                                      "note": "strong-scaling base of the --gpus N > 1 lines (same workload, one GPU)"}
        s5.close()

    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(budget_s=args.cpu_budget)
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
