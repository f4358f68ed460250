#!/usr/bin/env python3
"""How many planes would each rank of a P-way z-slab split of config 4 have to fetch for one advection?  Runs the
workload on ONE GPU for a few steps, then evaluates, on the host, the global-reach rule (every rank fetches max|u_z| *
dt * D + 2 planes on either side) against exact per-rank windows (the planes its own cells' back-traces touch).
python tools/gather_window_study.py [workload] [P] [steps]"""
import json
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fluid_simulation_amd as F  # noqa: E402
from bench import WORKLOADS, add_obstacles  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c4"
P = int(sys.argv[2]) if len(sys.argv) > 2 else 8
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
cfg = WORKLOADS[name]
W, H, D = cfg["W"], cfg["H"], cfg["D"]
sim = F.Simulation(W, H, D, steps, acc=cfg["acc"], quiet=1, dump_every=0)
with tempfile.TemporaryDirectory() as tmp:
    add_obstacles(F, sim, cfg, tmp)
out = {"workload": name, "ranks": P, "grid": [W, H, D], "per_step": []}
kz = np.float32(0.05) * np.float32(D)
dl = D // P
for s in range(steps):
    sim.run_one()
    if s < steps - 3:
        continue
    vz = sim.get(F.VZ)[1:-1, 1:-1, 1:-1]                       # (D, H, W)
    zg = np.arange(1, D + 1, dtype=np.float32)[:, None, None]
    pz = np.clip(zg - kz * vz, np.float32(0.5), np.float32(D) + np.float32(0.5))
    z0 = np.floor(pz).astype(np.int32)
    umax = float(np.abs(vz).max())
    reach = min(D, int(np.ceil(abs(float(kz)) * umax) + 2))
    old = new = 0
    per_rank = []
    for r in range(P):
        lo_own, hi_own = r * dl + 1, r * dl + dl
        zz = z0[r * dl:(r + 1) * dl]
        lo, hi = int(zz.min()), int(zz.max()) + 1
        need_new = max(0, lo_own - lo) + max(0, hi - hi_own)
        need_old = (min(reach, lo_own - 0)) + (min(reach, D + 1 - hi_own))
        per_rank.append([lo_own - lo if lo < lo_own else 0, hi - hi_own if hi > hi_own else 0])
        old += need_old
        new += need_new
    out["per_step"].append({"step": s + 1, "max_abs_vz": umax, "global_reach_planes": reach, "planes_fetched_global_rule": old,
                            "planes_fetched_exact_windows": new, "per_rank_below_above": per_rank})
print(json.dumps(out))
