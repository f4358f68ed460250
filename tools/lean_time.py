#!/usr/bin/env python3
"""One box, one process: the three-sweep kernel as shipped against the "lean interior" experiment (sweep_abl=16: 16-row
bands, 16 waves, no right-hand-side ring in LDS, wall-free body only, EVERY workgroup -- results wrong at the walls, timing
only) at 5 / 6 / 4 z chunks per band (215 / 258 / 172 workgroups).  python tools/lean_time.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fluid_simulation_amd as F  # noqa: E402

W = H = D = 512
sim = F.Simulation(W, H, D, 1, acc=6, quiet=1, dump_every=0)
sim.addObstacle(W // 3, H // 2, D // 2)
sim.run_one()
sim.run_one()
sim.sync()
out = {"grid": [W, H, D], "shipped_plan": sim._geti("triple_plan")}
shipped = "-1,%d" % sim._geti("triple_plan")
for rep in range(3):
    for name, abl, plans in (("shipped", 0, shipped), ("lean_5_chunks_215_wg", 16, "-1,0"), ("lean_6_chunks_258_wg", 16, "-1,8"),
                             ("lean_4_chunks_172_wg", 16, "-1,16")):
        sim.set_option("sweep_abl", abl)
        sim.set_option("launch_plans", plans)
        ms = sorted(sim.time_sweeps(0, F.PRESSURE, F.DIVERGENCE, 1.0, 6.0, 42) for _ in range(5))
        out.setdefault(name + "_ms_per_sweep", []).append(round(ms[0], 5))
print(json.dumps(out))
