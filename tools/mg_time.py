#!/usr/bin/env python3
"""solver=mg against solver=jacobi on the benchmark workloads, one box, one process: ms per step, ms per projection
family, and the relative residual of the pressure equation each leaves (simulation.cpp:263-271's fixed point).
python tools/mg_time.py [c2|c3|c4] [cycles | 0 = 2, 4 and 8] [fp32|fp64]"""
import json
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fluid_simulation_amd as F  # noqa: E402
from bench import WORKLOADS, add_obstacles  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c2"
cycles = int(sys.argv[2]) if len(sys.argv) > 2 and int(sys.argv[2]) > 0 else 0
prec = sys.argv[3] if len(sys.argv) > 3 else "fp32"
cfg = WORKLOADS[name]
W, H, D, acc = cfg["W"], cfg["H"], cfg["D"], cfg["acc"]


def residual(sim):
    p = sim.get(F.PRESSURE).astype(np.float64)
    div = sim.get(F.DIVERGENCE).astype(np.float64)
    solid = sim.get(F.OBS) == 1
    nb = p[1:-1, 1:-1, 2:] + p[1:-1, 1:-1, :-2] + p[1:-1, 2:, 1:-1] + p[1:-1, :-2, 1:-1] + p[2:, 1:-1, 1:-1] + p[:-2, 1:-1, 1:-1]
    r = div[1:-1, 1:-1, 1:-1] + nb - 6.0 * p[1:-1, 1:-1, 1:-1]
    live = ~solid[1:-1, 1:-1, 1:-1]
    return float(np.linalg.norm(r[live]) / np.linalg.norm(div[1:-1, 1:-1, 1:-1][live]))


out = {"workload": name, "grid": [W, H, D], "acc": acc, "precision": prec}
runs = [("jacobi", 0)] + [("mg", c) for c in ([cycles] if cycles else [2, 4, 8])]
for solver, cyc in runs:
    sim = F.Simulation(W, H, D, 1, acc=acc, solver=solver, quiet=1, dump_every=0, profile=1, mg_cycles=max(cyc, 1), precision=prec)
    with tempfile.TemporaryDirectory() as tmp:
        add_obstacles(F, sim, cfg, tmp)
    for _ in range(3):
        sim.run_one()
    sim.sync()
    sim.reset_timing()
    t0 = time.perf_counter()
    for _ in range(10):
        sim.run_one()
    sim.sync()
    dt = (time.perf_counter() - t0) / 10
    fam = {k: round(sim.timing(k)[0] / 10, 4) for k in ("sweep", "sweep_pair", "sweep_triple", "multigrid", "divergence", "gradient", "advect")}
    # one more projection of the developed flow, timed on its own, and the residual it leaves
    sim.sync()
    t0 = time.perf_counter()
    sim.project()
    sim.sync()
    proj = (time.perf_counter() - t0) * 1e3
    key = solver if solver == "jacobi" else "mg_%d_cycles" % cyc
    out[key] = {"ms_per_step": round(dt * 1e3, 3), "kernel_ms_per_step": fam, "one_projection_ms": round(proj, 3),
                "pressure_relative_residual": residual(sim), "mg_levels": sim._geti("mg_levels")}
    sim.close()
print(json.dumps(out))
