#!/usr/bin/env python3
"""One box, one process: the three-sweep kernel with one general body for all workgroups (wall_free=0) and with the
wall-free second body for interior workgroups (wall_free=1).  python tools/wall_free_time.py [W H D]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fluid_simulation_amd as F  # noqa: E402

W, H, D = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (512, 512, 512)
sim = F.Simulation(W, H, D, 1, acc=6, quiet=1, dump_every=0)
sim.addObstacle(W // 3, H // 2, D // 2)
sim.run_one()
sim.run_one()
sim.sync()
out = {"grid": [W, H, D]}
for rep in range(3):
    for mode in (0, 1):
        sim.set_option("wall_free", mode)
        ms = sorted(sim.time_sweeps(0, F.PRESSURE, F.DIVERGENCE, 1.0, 6.0, 42) for _ in range(5))
        out.setdefault("wall_free=%d" % mode, []).append(round(ms[0], 5))
out["triple_plan"] = sim._geti("triple_plan")
print(json.dumps(out))
