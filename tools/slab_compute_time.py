#!/usr/bin/env python3
"""Compute-side time of ONE z-slab of a P-way split, with the exchanges skipped ("FSNULL:" id):
what a rank spends in kernels (incl. the boundary-first split) when communication is free.
Fields are garbage across slab boundaries; only the timing means anything (development tool).
python tools/slab_compute_time.py [W H D] [P] [rank] [acc] [steps]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fluid_simulation_amd as F  # noqa: E402

W, H, D = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (1024, 512, 512)
P = int(sys.argv[4]) if len(sys.argv) > 4 else 8
rank = int(sys.argv[5]) if len(sys.argv) > 5 else P // 2
acc = int(sys.argv[6]) if len(sys.argv) > 6 else 80
steps = int(sys.argv[7]) if len(sys.argv) > 7 else 3
out = {"grid": [W, H, D], "ranks": P, "rank": rank, "acc": acc}
for overlap in (1, 2, 0):
    sim = F.Simulation(W, H, D, steps, acc=acc, quiet=1, dump_every=0, profile=1, overlap=overlap)
    if P > 1:
        sim.comm_init(rank, P, b"FSNULL:".ljust(128, b"\0"))
    sim.addObstacle(W // 3, H // 2, min(D, sim.z_offset + 2))
    sim.run_one()
    sim.sync()
    sim.reset_timing()
    t0 = time.perf_counter()
    for _ in range(steps):
        sim.run_one()
    sim.sync()
    dt = (time.perf_counter() - t0) / steps
    fam = {k: sim.timing(k) for k in ("sweep", "sweep_pair", "divergence", "gradient", "advect", "comm", "misc")}
    out["overlap=%d" % overlap] = {"ms_per_step": dt * 1e3, "cells_steps_per_sec_if_all_ranks_alike": W * H * D / dt,
                                   "kernel_ms_per_step": {k: v[0] / steps for k, v in fam.items()},
                                   "launches_per_step": {k: v[1] / steps for k, v in fam.items()}}
    sim.close()
    if P == 1:
        break
print(json.dumps(out))
