#!/usr/bin/env python3
"""Runs a handful of solver sweeps (and nothing else heavy) so that rocprofv3 --pmc output is
easy to read: python tools/pmc_sweep.py [W H D] [reps] (development tool)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fluid_simulation_amd as F  # noqa: E402

W, H, D = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (512, 512, 512)
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 6
sim = F.Simulation(W, H, D, 1, acc=1, quiet=1, dump_every=0)
for k, v in (("sweep_ry", os.environ.get("RY", "4")), ("sweep_zc", os.environ.get("ZC", "0")),
             ("sweep_abl", os.environ.get("ABL", "0"))):
    sim.set_option(k, v)
sim.addObstacle(W // 3, H // 2, D // 2)
a = 0.05 * 2.0e-5 * W * H * D
ms = sim.time_sweeps(2, F.VY, F.VY_PREV, a, 1 + 6 * a, reps)
print("sweep %.1f us" % (ms * 1e3))
