#!/usr/bin/env python3
"""Timing of the advection kernels in ONE process on ONE box (box-to-box spread is ~3 %), each variant on its own fresh
simulation run exactly like bench.py's (same obstacles, 3 warm-up steps, then 20 timed steps), because the cost of the
gathers follows the flow as it develops:
python tools/advect_time.py [c2|c3|c4] [fp32|fp64] [acc] [solver] -> ms per step spent in the advect family for the row kernels,
the per-cell kernels with clamp tables, and the plain per-cell kernels; steps 4-23 and steps 24-43."""
import json
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fluid_simulation_amd as F  # noqa: E402
from bench import WORKLOADS, add_obstacles  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c3"
prec = sys.argv[2] if len(sys.argv) > 2 else "fp32"
cfg = WORKLOADS[name]
acc = int(sys.argv[3]) if len(sys.argv) > 3 else cfg["acc"]
solver = sys.argv[4] if len(sys.argv) > 4 else "jacobi"
out = {"workload": name, "precision": prec, "acc": acc, "solver": solver}
for rep in range(2):
    for kind in ("row", "celltab", "cell", "tile"):
        sim = F.Simulation(cfg["W"], cfg["H"], cfg["D"], 1, acc=acc, precision=prec, quiet=1, dump_every=0, profile=1,
                           advect_kernels=kind, solver=solver)
        with tempfile.TemporaryDirectory() as tmp:
            add_obstacles(F, sim, cfg, tmp)
        for _ in range(3):
            sim.run_one()
        res = []
        for _ in range(2):
            sim.sync()
            sim.reset_timing()
            for _ in range(20):
                sim.run_one()
            sim.sync()
            ms, n = sim.timing("advect")
            res.append(round(ms / 20, 4))
        out["%s_%d" % (kind, rep)] = {"advect_ms_per_step_steps_4_23": res[0], "steps_24_43": res[1]}
        sim.close()
print(json.dumps(out))
