#!/usr/bin/env python3
"""BASELINE config 5: the same tunnel in fp32 and fp64 fields -- field differences (tolerance) and
time per step (bandwidth sensitivity).  python tools/fp64_vs_fp32.py [N] [steps] [acc]"""
import json
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fluid_simulation_amd as F  # noqa: E402
from fluid_simulation_amd import shapes  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
acc = int(sys.argv[3]) if len(sys.argv) > 3 else 80
res = {"grid": [N, N, N], "steps": steps, "acc": acc}
fields = {}
with tempfile.TemporaryDirectory() as tmp:
    sphere = shapes.write_binary_stl(os.path.join(tmp, "s.stl"), shapes.sphere_triangles(2.0, 48, 24))
    plate = shapes.write_binary_stl(os.path.join(tmp, "p.stl"), shapes.box_triangles(0.2, 2.4, 1.6))
    for prec in ("fp32", "fp64"):
        sim = F.Simulation(N, N, N, steps, acc=acc, quiet=1, dump_every=0, precision=prec)
        F.loadSTLIntoObstacles(sphere, sim, 0.3, 0.0, 0.0, 0.0, -N / 4.0, 0.0, 0.0)
        F.loadSTLIntoObstacles(plate, sim, 0.45, 0.0, 0.0, 0.0, N / 8.0, 0.0, 0.0)
        sim.run_one()
        sim.sync()
        t0 = time.perf_counter()
        for _ in range(steps - 1):
            sim.run_one()
        sim.sync()
        res[prec + "_ms_per_step"] = (time.perf_counter() - t0) / max(1, steps - 1) * 1e3
        fields[prec] = {n: sim.get(f, dtype=np.float64) for f, n in ((F.DENS, "dens"), (F.VX, "v_x"), (F.VY, "v_y"),
                                                                       (F.VZ, "v_z"), (F.PRESSURE, "pressure"))}
        sim.close()
res["rel_l2_fp32_vs_fp64"] = {}
for n in fields["fp32"]:
    a, b = fields["fp32"][n], fields["fp64"][n]
    res["rel_l2_fp32_vs_fp64"][n] = float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-300))
res["time_ratio_fp64_over_fp32"] = res["fp64_ms_per_step"] / res["fp32_ms_per_step"]
print(json.dumps(res))
