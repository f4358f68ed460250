#!/usr/bin/env python3
"""Sweep-kernel tuning / ablation table on one GPU (development tool, not a test).

    python tools/tune_sweep.py [W H D]

Times fs_time_sweeps (HIP events on the solver stream) for launch-shape variants and for
timing-only ablation builds that drop one stream each (results of those are wrong by design).
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import fluid_simulation_amd as F  # noqa: E402

W, H, D = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (512, 512, 512)
reps = int(os.environ.get("REPS", "40"))
sim = F.Simulation(W, H, D, 1, acc=1, quiet=1, dump_every=0)
sim.addObstacle(W // 3, H // 2, D // 2)
sim.run_one()
sim.sync()
cells = W * H * D
a = 0.05 * 2.0e-5 * W * H * D
rows = []


def run(tag, **opts):
    for k, v in opts.items():
        sim.set_option(k, v)
    ms = min(sim.time_sweeps(2, F.VY, F.VY_PREV, a, 1 + 6 * a, reps) for _ in range(3))
    rows.append(dict(tag=tag, ms=ms, GBps_alg=12 * cells / ms / 1e6, **opts))
    print("%-28s %8.1f us   %7.0f GB/s algorithmic" % (tag, ms * 1e3, 12 * cells / ms / 1e6), flush=True)


base = dict(sweep_ry=4, sweep_zc=0, sweep_blocks=2048, sweep_abl=0)
run("default ry4 auto", **base)
for ry in (2, 4):
    for zc in (8, 16, 32, 64, 128):
        run("ry%d zc%d" % (ry, zc), **dict(base, sweep_ry=ry, sweep_zc=zc))
for abl, name in ((1, "no flags"), (2, "no y-halo rows"), (4, "no rhs"), (8, "no store"), (3, "no flags+halo")):
    run("ablation: " + name, **dict(base, sweep_abl=abl, sweep_zc=32))
with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "tune_sweep.json"), "w") as f:
    json.dump(rows, f, indent=1)
