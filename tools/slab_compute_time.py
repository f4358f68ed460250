#!/usr/bin/env python3
"""What one z-slab of a P-way split costs, two ways (development tool, one GPU):

  FSNULL (default)  ONE rank process with the exchanges skipped: the compute side of a slab step (incl. the
                    boundary-first split) when communication is free.  Fields are garbage across slab
                    boundaries; only the timing means anything.
  ipc               P rank processes that SHARE this GPU and exchange over the FSIPC transport (device-to-device
                    copies between the processes, stream-ordered handshakes): the communication schedules really run
                    asynchronously against peers.  P ranks on one GPU take about P times a rank's time, so this is
                    not a scaling number either -- it shows what each schedule costs when the exchange is real.

python tools/slab_compute_time.py [W H D] [P] [rank] [acc] [steps] [null|ipc]"""
import json
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fluid_simulation_amd as F  # noqa: E402

W, H, D = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (1024, 512, 512)
P = int(sys.argv[4]) if len(sys.argv) > 4 else 8
rank = int(sys.argv[5]) if len(sys.argv) > 5 else P // 2
acc = int(sys.argv[6]) if len(sys.argv) > 6 else 80
steps = int(sys.argv[7]) if len(sys.argv) > 7 else 3
mode = sys.argv[8] if len(sys.argv) > 8 else "null"
FAMS = ("sweep", "sweep_pair", "sweep_triple", "divergence", "gradient", "advect", "comm", "misc", "multigrid")


def one_rank(uid, overlap):
    sim = F.Simulation(W, H, D, steps, acc=acc, quiet=1, dump_every=0, profile=1, overlap=overlap,
                       solver=os.environ.get("FS_SOLVER", "jacobi"),      # FS_SOLVER=mg: the multigrid pressure solve on slabs
                       mg_min_planes=int(os.environ.get("FS_MG_MIN_PLANES", "4")))
    if P > 1:
        sim.comm_init(rank, P, uid)
    sim.addObstacle(W // 3, H // 2, D // 2)              # every rank issues the same call (rank-symmetric bookkeeping)
    sim.run_one()
    sim.sync()
    sim.reset_timing()
    t0 = time.perf_counter()
    for _ in range(steps):
        sim.run_one()
    sim.sync()
    dt = (time.perf_counter() - t0) / steps
    fam = {k: sim.timing(k) for k in FAMS}
    res = {"ms_per_step": dt * 1e3, "overlap_plan": sim._geti("overlap_plan"),
           "kernel_ms_per_step": {k: v[0] / steps for k, v in fam.items()},
           "launches_per_step": {k: v[1] / steps for k, v in fam.items()},
           "stream_syncs": sim._geti("stream_syncs"), "reach_hidden": sim._geti("reach_hidden"),
           "reach_exposed": sim._geti("reach_exposed")}
    sim.close()
    return res


if mode == "child":                                      # one rank of an ipc run: argv[9] = id file, argv[10] = overlap
    uid = open(sys.argv[9], "rb").read()
    print(json.dumps(one_rank(uid, sys.argv[10])))
    sys.exit(0)

out = {"grid": [W, H, D], "ranks": P, "acc": acc, "transport": mode, "solver": os.environ.get("FS_SOLVER", "jacobi")}
for overlap in ("1", "2", "0", "3", "auto"):
    if mode == "ipc":
        idfile = "/tmp/fs_slab_time_%d.id" % os.getpid()
        open(idfile, "wb").write(F.comm_unique_id("ipc"))
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), str(W), str(H), str(D), str(P), str(r), str(acc), str(steps),
                                   "child", idfile, overlap], stdout=subprocess.PIPE, text=True) for r in range(P)]
        res = [json.loads(p.communicate(timeout=900)[0].strip().splitlines()[-1]) for p in procs]
        out["overlap=%s" % overlap] = {"ms_per_step_slowest_rank": max(r["ms_per_step"] for r in res),
                                       "note": "%d ranks share ONE GPU" % P, "per_rank": res}
    else:
        out["rank"] = rank
        r = one_rank((b"FSNULL:push" if overlap == "3" else b"FSNULL:").ljust(128, b"\0"), overlap)   # 3: the push kernels, storing into the rank's own halos
        r["cells_steps_per_sec_if_all_ranks_alike"] = W * H * D / (r["ms_per_step"] * 1e-3)
        out["overlap=%s" % overlap] = r
    if P == 1:
        break
print(json.dumps(out))
