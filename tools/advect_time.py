#!/usr/bin/env python3
"""A/B timing of the advection kernels in ONE process on ONE box (box-to-box spread is ~3 %):
python tools/advect_time.py [c2|c3|c4] [fp32|fp64]  -> ms per step spent in the advect family, row vs cell kernels."""
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
sim = F.Simulation(cfg["W"], cfg["H"], cfg["D"], 1, acc=4, precision=prec, quiet=1, dump_every=0, profile=1)
with tempfile.TemporaryDirectory() as tmp:
    add_obstacles(F, sim, cfg, tmp)
for _ in range(6):
    sim.run_one()
out = {"workload": name, "precision": prec, "acc": 4}
for rep in range(2):
    for kind in ("row", "cell"):
        sim.set_option("advect_kernels", kind)
        sim.run_one()
        sim.sync()
        sim.reset_timing()
        for _ in range(5):
            sim.run_one()
        sim.sync()
        ms, n = sim.timing("advect")
        out["%s_%d" % (kind, rep)] = {"advect_ms_per_step": ms / 5, "launches_per_step": n / 5}
print(json.dumps(out))
