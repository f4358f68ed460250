#!/usr/bin/env python3
"""Pair and triple (temporal-blocked) sweep kernels vs single sweeps (development tool)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fluid_simulation_amd as F  # noqa: E402

W, H, D = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (512, 512, 512)
reps = int(os.environ.get("REPS", "40"))
sim = F.Simulation(W, H, D, 1, acc=1, quiet=1, dump_every=0)
sim.addObstacle(W // 3, H // 2, D // 2)
sim.run_one()
sim.sync()
cells = W * H * D
a = 0.05 * 2.0e-5 * W * H * D


def run(tag, **opts):
    for k, v in opts.items():
        sim.set_option(k, v)
    ms = min(sim.time_sweeps(2, F.VY, F.VY_PREV, a, 1 + 6 * a, reps) for _ in range(3))
    print("%-34s %8.1f us/sweep  %7.0f GB/s algorithmic  (%.1f%% of 8 TB/s)" % (
        tag, ms * 1e3, 12 * cells / ms / 1e6, 12 * cells / ms / 1e6 / 80), flush=True)


run("single ry2 zc32", sweep_fuse=1, sweep_ry=2, sweep_zc=32)
for small in (0, 2, 1):
    for zc in (0, 32, 43, 52, 64, 86, 128, 171, 256):
        run("pair shape=%d zc=%d" % (small, zc), sweep_fuse=2, pair_shape=small, pair_zc=zc)
run("three sweeps per pass (tuned plan)", sweep_fuse=4, pair_shape=0, pair_zc=0)
sim.set_option("sweep_fuse", "3")
