"""One rank of a z-slab run (spawned by tests/test_gpu_slabs.py).  argv: rank nranks idfile outdir W H D acc steps"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fluid_simulation_amd as F  # noqa: E402


def ball(W, H, D, cx, cy, cz, r):
    z, y, x = np.mgrid[0:D + 2, 0:H + 2, 0:W + 2]
    m = ((x - cx) ** 2 + (y - cy) ** 2 + (z - cz) ** 2) <= r * r
    m[0] = m[-1] = False
    m[:, 0] = m[:, -1] = False
    m[:, :, 0] = m[:, :, -1] = False
    return m


def main():
    rank, nranks = int(sys.argv[1]), int(sys.argv[2])
    idfile, outdir = sys.argv[3], sys.argv[4]
    W, H, D, acc, steps = (int(v) for v in sys.argv[5:10])
    stl = sys.argv[10] if len(sys.argv) > 10 else ""
    precision = sys.argv[11] if len(sys.argv) > 11 else "fp32"
    solver = sys.argv[12] if len(sys.argv) > 12 else "jacobi"
    extra = {"sor_omega": 1.6} if solver == "rbsor" else {}
    for kv in (sys.argv[13].split(",") if len(sys.argv) > 13 and sys.argv[13] else []):   # further options, k=v,k=v
        k, v = kv.split("=")
        extra[k] = v
    kw = dict(acc=acc, quiet=1, dump_dir=os.path.join(outdir, "data"), dump_every=1, voxel_seed=77, debug_poison_gather=1,
              precision=precision, solver=solver)
    kw.update(extra)                                     # options of the test case win (e.g. dump_every=0 at full size)
    sim = F.Simulation(W, H, D, steps, **kw)
    if nranks > 1:
        sim.comm_init(rank, nranks, open(idfile, "rb").read())
    Dl, zoff = sim.local_depth, sim.z_offset
    gmask = ball(W, H, D, W / 3.0, H / 2.0, D / 2.0 + 1.5, min(W, H, D) / 4.0)   # straddles slab boundaries
    gmask[D // 2, 2, 2] = gmask[D // 2 + 1, 2, 2] = True
    sim.set_mask(gmask[zoff:zoff + Dl + 2])
    if stl:
        F.loadSTLIntoObstacles(stl, sim, 0.5, 0.0, 20.0, 0.0, 3.0, 0.0, 0.0)
    sim.run()
    # what the slab steps of that run cost the host: stream synchronisations (none: the reach of every back-trace arrives
    # asynchronously), waits for a reach, those that found it not yet delivered; and the communication schedule in force
    sched = np.array([sim._geti(k) for k in ("stream_syncs", "reach_waits", "reach_waits_blocked", "overlap_plan", "comm_cus_plan")])
    # host-side edits on and next to slab boundaries (every rank issues the same calls; a rank applies
    # those that fall into its planes), then two more steps: the stale-halo bookkeeping must catch them
    for zb in sorted({D // 2, D // 2 + 1, max(1, D // 4), min(D, 3 * D // 4 + 1)}):
        sim.addDensity(5, 4, zb, 0.25)
        sim.setVelocity(6, 5, zb, 1.5, -0.5, 2.0)
        sim.addObstacle(9, 7, zb)
    sim.set_option("dump_every", 0)
    sim.run_one()
    sim.run_one()
    # edits that fall into ONE rank's planes only -- an interior plane, then a slab-boundary plane whose
    # neighbour holds a halo copy: the dirty-halo bookkeeping must stay rank-symmetric (collectives pair up)
    for zb in (2, D // 2, D // 2 + 1, max(1, D // 3)):
        sim.addDensity(7, 3, zb, 0.5)
        sim.setVelocity(4, 6, zb, -1.0, 0.75, 0.5)
        sim.run_one()
    sim.addObstacle(11, 5, D // 2)
    sim.run_one()
    out = {F.FIELD_NAMES[f]: sim.get(f) for f in (F.DENS, F.VX, F.VY, F.VZ, F.OBS, F.PRESSURE)}
    if os.environ.get("FS_SLAB_DIGEST"):
        # full-size grids: one SHA-256 per plane instead of gigabytes of arrays (equality of digests = equality of bits)
        import hashlib
        out = {k: np.frombuffer(b"".join(hashlib.sha256(np.ascontiguousarray(pl).tobytes()).digest() for pl in a),
                                dtype=np.uint8).reshape(a.shape[0], 32) for k, a in out.items()}
    stats = np.array(sim.stats(F.DENS) + sim.stats(F.VX))
    reach = sim._geti("last_advect_reach")
    kernels = np.array([sim._geti("triple_plan"), sim._geti("two_sweep_fused"), sim._geti("halo_depth")])
    sched_end = np.array([sim._geti(k) for k in ("stream_syncs", "reach_waits", "reach_waits_blocked")])
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), zoff=zoff, stats=stats, reach=reach, kernels=kernels, sched=sched,
             sched_end=sched_end, **out)
    sim.close()


if __name__ == "__main__":
    main()
